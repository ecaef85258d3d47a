/* orbx.h -- C ABI of the MI355X-native ORB-SLAM2 hot path (liborbx.so).
 *
 * This is the drop-in boundary.  The reference has no plugin
 * registry: its "operator interface" for this path is three C++ class surfaces,
 *   ORB_SLAM2::ORBextractor   /root/reference/include/ORBextractor.h:92-161
 *   ORB_SLAM2::ORBmatcher     /root/reference/include/ORBmatcher.h:57-215
 *   ORB_SLAM2::Optimizer      /root/reference/include/Optimizer.h:112
 * The header-compatible C++ classes in self_commit_orb-slam2_amd/shim/ keep those
 * surfaces and marshal to the functions declared here; INTEGRATION.md shows the
 * binding.  Plain pointers and sizes only; no C++ / torch types.
 *
 * Conventions: every function returns ORBX_OK (0) or a negative ORBX_ERR_* code and
 * never throws; orbx_last_error() returns a thread-local message for the last
 * failure.  There is NO CPU fallback: without a usable HIP device the create
 * functions fail with ORBX_ERR_NODEVICE.  A handle owns one HIP device + stream and
 * its scratch memory; a handle is not re-entrant (like an ORBextractor instance,
 * ORBextractor.h:161) but different handles may be used from different threads
 * (the stereo Frame constructor does exactly that, src/Frame.cc:159-167).
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_OK 0
#define ORBX_ERR_ARG (-1)      /* bad argument                                  */
#define ORBX_ERR_HIP (-2)      /* HIP runtime / kernel failure                  */
#define ORBX_ERR_CAPACITY (-3) /* an internal or caller buffer was too small    */
#define ORBX_ERR_NODEVICE (-4) /* no usable gfx950 device                       */
#define ORBX_ERR_STATE (-5)    /* call sequence error (e.g. results before run) */

const char *orbx_last_error(void);
/* Library/ABI version: major*10000 + minor*100 + patch. */
int orbx_version(void);
/* "<uuid>@<pci address>" of HIP device `device` into identity[capacity >= 64] and the NUMA node of its PCI function (-1 = unknown; may be
 * NULL): what a multi-process run uses to prove that its N ranks sit on N distinct GPUs (bench.py).  Creates no stream. */
int orbx_device_identity(int device, char *identity, int capacity, int *numa_node);

/* ------------------------------------------------------------------------------------
 * ORB extractor  ==  ORB_SLAM2::ORBextractor
 * ---------------------------------------------------------------------------------- */
typedef struct orbx_extractor orbx_extractor;

/* == cv::KeyPoint, 28 bytes (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct orbx_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbx_keypoint;

typedef struct orbx_extractor_config {
    /* the five ORBextractor constructor arguments, ORBextractor.h:92 /
     * src/ORBextractor.cc:492-496 (values come from the YAML, src/Tracking.cc:168-192) */
    int nfeatures;
    float scale_factor;
    int nlevels;
    int ini_th_fast;
    int min_th_fast;
    /* sizing of the handle's device buffers */
    int max_width, max_height; /* largest image accepted                          */
    int max_batch;             /* most frames per orbx_extract_batch* call (>=1)  */
    int device;                /* HIP device ordinal                              */
    /* 7-tap fixed-point Gaussian (sum 256) used for cv::GaussianBlur(7x7, sigma 2),
     * src/ORBextractor.cc:1629.  All zero selects the OpenCV 4.x taps
     * 18,34,48,56,48,34,18. */
    uint16_t gauss_taps[7];
    uint16_t reserved_;
} orbx_extractor_config;

/* ORBextractor::ORBextractor (src/ORBextractor.cc:492-609): scale tables, per-level
 * quotas, pattern and umax are built here, device buffers are allocated. */
int orbx_extractor_create(const orbx_extractor_config *cfg, orbx_extractor **out);
void orbx_extractor_destroy(orbx_extractor *h);

/* GetLevels / GetScaleFactor(s) / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (ORBextractor.h:118-158) + mnFeaturesPerLevel.
 * Any pointer may be NULL; arrays hold nlevels entries. */
int orbx_extractor_tables(const orbx_extractor *h, int *nlevels, float *scale, float *inv_scale,
                          float *sigma2, float *inv_sigma2, int *features_per_level);
/* Upper bound of keypoints one frame can yield: sum over levels of quota+3
 * (octree exit conditions, src/ORBextractor.cc:910,1003). */
/* The same tables without a handle (and without a device): what ORBextractor::ORBextractor computes from
 * (nfeatures, scaleFactor, nlevels) alone (src/ORBextractor.cc:499-554).  Only cfg->nfeatures / scale_factor / nlevels are read.
 * The drop-in constructor uses it so that the getters are valid even when no HIP device can be opened (the reference's
 * constructor cannot fail). */
int orbx_extractor_tables_for(const orbx_extractor_config *cfg, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                              int *features_per_level);
int orbx_extractor_capacity(const orbx_extractor *h);

/* ORBextractor::operator() for one host image WITHOUT the copy into caller arrays: *keypoints / *descriptors point into the
 * handle's pinned result buffer (count entries / count*32 bytes), valid until the next call on this handle.  The drop-in functor
 * (shim/ORBextractor.cc) converts to cv::KeyPoint / cv::Mat straight from there.  Empty image: ORBX_OK, *count = 0. */
int orbx_extract_view(orbx_extractor *h, const uint8_t *image, int width, int height, int stride, const orbx_keypoint **keypoints,
                      const uint8_t **descriptors, int *count);

/* The same call, plus the host copy of the image pyramid the reference keeps in the public member mvImagePyramid
 * (include/ORBextractor.h:161; read by Frame::ComputeStereoMatches, src/Frame.cc:1044,1248,1272,1281): `pyramid` (may be NULL = not
 * wanted) receives VIEWS into the handle's pinned memory - level 0 is the handle's staged copy of the caller's image, levels >= 1
 * arrive with the results of the same launch set (no second transfer, no second wait, no host copy) -, valid until the next call on
 * this handle.  Rows of level l are stride[l] bytes apart. */
#define ORBX_PYRAMID_MAX_LEVELS 12
typedef struct orbx_host_pyramid {
    const uint8_t *level[ORBX_PYRAMID_MAX_LEVELS];
    int width[ORBX_PYRAMID_MAX_LEVELS], height[ORBX_PYRAMID_MAX_LEVELS], stride[ORBX_PYRAMID_MAX_LEVELS];
    int nlevels;
} orbx_host_pyramid;
int orbx_extract_view_pyramid(orbx_extractor *h, const uint8_t *image, int width, int height, int stride, const orbx_keypoint **keypoints,
                              const uint8_t **descriptors, int *count, orbx_host_pyramid *pyramid);

/* Single-frame calls (orbx_extract_view*, orbx_extract, orbx_extract_batch with one frame on a max_batch = 1 handle) that are inside the
 * library at the same moment - from different handles on different threads, same device / configuration / image size - are COMBINED
 * into one launch set on a shared engine (csrc/orbx_extractor.hip: "the combiner"); each call returns exactly what it would have
 * returned alone.  A lone caller never waits for company.  ORBX_COMBINE=0 in the environment gives every handle its own graph instead;
 * ORBX_COMBINE_MAX (16) = most frames per set, ORBX_COMBINE_ENGINES (2) = sets in flight.
 * orbx_extractor_expect_partner: a one-shot hint for the NEXT call on `h` - `partner`'s call is about to arrive (the other extractor
 * thread of the stereo Frame constructor, src/Frame.cc:159-167): with ORBX_COMBINE_PARTNER_US=<us> in the environment the set waits that
 * long for it instead of leaving without it.  Off by default: measured on the reference's constructor, the 30-40 us between its two
 * thread starts cost more than the second launch set saves (the second call simply takes the other engine).
 * orbx_combiner_stats: launch sets and frames served so far for h's configuration (frames / batches = mean set size). */
int orbx_extractor_expect_partner(orbx_extractor *h, orbx_extractor *partner);
int orbx_combiner_stats(const orbx_extractor *h, int64_t *batches, int64_t *frames, int *engines);
/* Microseconds summed so far over h's configuration: us4[0] staging copies (per call), [1] leaders' wait for an engine / for company,
 * [2] graph launch calls, [3] device time + synchronisation (per launch set).  A measurement aid (tools/latency_shim.py). */
int orbx_combiner_profile(const orbx_extractor *h, double *us4);
int orbx_combiner_reset_stats(orbx_extractor *h);
int orbx_combiner_histogram(const orbx_extractor *h, int maxn, int64_t *sets, double *mean_us);   /* sets[n], mean_us[n] for n = 0..maxn frames per launch set */

/* ORBextractor::operator() (ORBextractor.h:110, src/ORBextractor.cc:1544-1668) for one
 * host image.  `keypoints` / `descriptors` hold `capacity` entries / capacity*32 bytes;
 * *count receives the real number (<= orbx_extractor_capacity()).  An empty image
 * (NULL / zero size) returns ORBX_OK with *count = 0 (reference: silent return). */
int orbx_extract(orbx_extractor *h, const uint8_t *image, int width, int height, int stride,
                 orbx_keypoint *keypoints, uint8_t *descriptors, int capacity, int *count);

/* Batched form: `batch` independent frames of identical size.  Frame f writes
 * keypoints[f*capacity ...], descriptors[f*capacity*32 ...], counts[f].
 * Equivalent to `batch` operator() calls; the frames are data-parallel on the GPU.
 * A batch of 2 x ORBX_HOST_BATCH_CHUNK frames (environment, default 64; 0 = never) or more runs as a pipeline over chunks (staging / upload of
 * chunk c+1 and read-back of chunk c-1 under the kernels of chunk c).  The caller's arrays receive the whole batch either way, but after a
 * CHUNKED call the handle's device buffers hold only its last chunk: every "last batch" device-side view below (orbx_batch_results_device,
 * orbx_batch_status_device, orbx_extractor_status, orbx_batch_download, orbx_download_pyramid*, the debug taps, and the matcher / frame-finish
 * entry points that read the extractor's last batch) then returns ORBX_ERR_STATE instead of indexing a chunk.  Callers that want the results on
 * the device use orbx_upload_frames + orbx_extract_batch_device (or set ORBX_HOST_BATCH_CHUNK=0). */
int orbx_extract_batch(orbx_extractor *h, const uint8_t *const *images, int batch, int width,
                       int height, int stride, orbx_keypoint *keypoints, uint8_t *descriptors,
                       int capacity, int *counts);

/* The same call as a two-deep pipeline (SURVEY.md 8b's batch entry point with host pointers; replaces a loop of operator() calls,
 * include/ORBextractor.h:110, src/Frame.cc:394).  _begin stages and uploads the frames (a batch whose frames ALL live in pinned or registered
 * host memory - every byte of every frame, checked per frame - is read in place; one pageable frame and the whole batch is staged), enqueues the batch's launch set and the read-back of its results, and returns without waiting; _end waits
 * for the OLDEST begun batch and fills the caller's arrays exactly like orbx_extract_batch.  Up to two batches may be begun before the first
 * _end: staging + upload of batch i+1 and the read-back of batch i-1 then run under the kernels of batch i.  The images of a batch must stay
 * valid until its _begin returns (pinned / registered images: until its _end returns).  A third _begin, or an _end with nothing begun,
 * returns ORBX_ERR_STATE.  orbx_extract_batch itself runs this pipeline over chunks of its batch (ORBX_HOST_BATCH_CHUNK frames, default 64). */
int orbx_extract_batch_begin(orbx_extractor *h, const uint8_t *const *images, int batch, int width, int height, int stride);
int orbx_extract_batch_end(orbx_extractor *h, orbx_keypoint *keypoints, uint8_t *descriptors, int capacity, int *counts);

/* Device-resident batch: images_dev points to DEVICE memory, frame f at
 * images_dev + f*frame_pitch (rows `stride` bytes apart); the buffer spans batch*frame_pitch
 * bytes.  Runs asynchronously on the handle's stream; results stay in handle-owned device
 * buffers until downloaded.  Rows padded to stride >= round_up(width,4)+12 (and frame_pitch >=
 * stride*height) let the pyramid kernel read aligned windows everywhere (faster); tight rows
 * are handled too. */
int orbx_extract_batch_device(orbx_extractor *h, const void *images_dev, int batch, int width,
                              int height, int stride, size_t frame_pitch);
/* Device pointers of the last batch's results: keypoints[f*cap + i], descriptors
 * [(f*cap + i)*32], counts[f]; *capacity = per-frame capacity of those arrays.  "Last batch" = the last orbx_extract_batch_device /
 * single-frame / un-chunked host call (ORBX_ERR_STATE after a chunked orbx_extract_batch, see there). */
int orbx_batch_results_device(orbx_extractor *h, const orbx_keypoint **keypoints_dev,
                              const uint8_t **descriptors_dev, const int32_t **counts_dev,
                              int *capacity);
/* Wait for the stream and copy the last batch's results to host arrays laid out as
 * in orbx_extract_batch. */
int orbx_batch_download(orbx_extractor *h, int batch, orbx_keypoint *keypoints, uint8_t *descriptors,
                        int capacity, int *counts);
/* Upload host frames into a handle-owned device staging area (returns its device
 * pointer, stride and frame pitch) so callers without their own device allocator
 * (bench.py, tests) can keep inputs resident in HBM. */
int orbx_upload_frames(orbx_extractor *h, const uint8_t *const *images, int batch, int width,
                       int height, int stride, const void **images_dev, int *dev_stride,
                       size_t *dev_frame_pitch);
int orbx_extractor_sync(orbx_extractor *h);
/* Capacity status of the last batch WITHOUT downloading it (a device-resident pipeline never calls orbx_batch_download, which is
 * where a host consumer learns about an overflow): waits for the stream, *bits = OR over the frames of
 *   2 = quadtree node list, 4 = level keypoint buffer (internal invariants: the list never exceeds the level's quota + 3; bit 1,
 *   the former per-level candidate limit, no longer exists - the quadtree's point arrays hold every candidate the detector can emit).
 * 0 = every frame is complete: always, unless an internal invariant is broken; a set bit means the results of that batch are NOT
 * the reference's.  orbx_batch_status_device: the same words on the device, status_dev[f] per frame and
 * status_dev[batch] for the whole batch.  Matcher / frame calls that are chained behind an extractor (`after` argument) pick
 * the batch word up on the device, and their own download calls return ORBX_ERR_CAPACITY when it is set. */
int orbx_extractor_status(orbx_extractor *h, int32_t *bits);
int orbx_batch_status_device(orbx_extractor *h, const int32_t **status_dev, int *batch);

/* std::vector<cv::Mat> mvImagePyramid (ORBextractor.h:161, read by
 * Frame::ComputeStereoMatches, src/Frame.cc:1044,1248,1272,1281): size and bytes of
 * pyramid level `level` of frame `frame` of the last call.  blurred=1 returns the
 * Gaussian-blurred copy the descriptors were sampled from (src/ORBextractor.cc:1626-1634). */
/* Every level of frame `frame` of the last call in one device->host transfer: dst[l] receives level l (rows dst_strides[l] bytes
 * apart), l = 0..nlevels-1.  This is what refills the public mvImagePyramid member in shim/ORBextractor.cc. */
int orbx_download_pyramid_all(orbx_extractor *h, int frame, uint8_t *const *dst, const int *dst_strides, int nlevels);
int orbx_pyramid_level_size(const orbx_extractor *h, int width, int height, int level, int *w, int *hgt);
int orbx_download_pyramid(orbx_extractor *h, int frame, int level, int blurred, uint8_t *dst, int dst_stride);

/* Stage taps for the parity tests: FAST score map of a level
 * (0 = not a corner at minThFAST), the per-level candidate list in vToDistributeKeys
 * order (src/ORBextractor.cc:1089-1157) packed as x | y<<12 | score<<24 relative to the
 * border window origin, and the per-level keypoints after DistributeOctTree + IC_Angle
 * in level coordinates (src/ORBextractor.cc:1167-1198). */
/* The fused detector keeps FAST scores on chip; enable the taps to also write the score map. */
int orbx_extractor_set_debug_taps(orbx_extractor *h, int enable);
int orbx_debug_download_scores(orbx_extractor *h, int frame, int level, uint8_t *dst, int dst_stride);
int orbx_debug_download_candidates(orbx_extractor *h, int frame, int level, uint32_t *packed, int cap, int *count);
int orbx_debug_download_level_keypoints(orbx_extractor *h, int frame, int level, orbx_keypoint *kps, int cap, int *count);

/* Kernel timing measured with HIP events on the handle's own stream: average milliseconds
 * per stage (order given by orbx_stage_name(i)) and their sum, over the batch calls issued
 * since profiling was enabled (at most the last 64).  Every call keeps its own event set,
 * so nothing is synchronised inside a timed region; reading waits for the stream. */
#define ORBX_MAX_STAGES 16
int orbx_extractor_set_profiling(orbx_extractor *h, int enable);
int orbx_extractor_last_timing(orbx_extractor *h, float *total_ms, float *stage_ms, int *nstages);
const char *orbx_stage_name(int stage);


/* ------------------------------------------------------------------------------------
 * ORB matcher  ==  the Hamming paths of ORB_SLAM2::ORBmatcher (+ the Hamming stage of
 * Frame::ComputeStereoMatches).  The reference walks KeyFrame / MapPoint / Frame objects;
 * the C ABI takes the same information as flat arrays (INTEGRATION.md shows the
 * marshalling the class shim does).
 * ---------------------------------------------------------------------------------- */
typedef struct orbx_matcher orbx_matcher;

/* ORBmatcher::DescriptorDistance (ORBmatcher.h:65, src/ORBmatcher.cc:1913-1933): 256-bit
 * Hamming distance of two 32-byte descriptors (host helper, used by the class shim). */
int orbx_descriptor_distance(const uint8_t *a, const uint8_t *b);

/* Features of `nframes` frames, frame f at index f*capacity.  Device OR host pointers,
 * depending on the function. */
typedef struct orbx_feature_set {
    const orbx_keypoint *keypoints; /* angle/x/y/octave are read (mvKeys / mvKeysUn)             */
    const uint8_t *descriptors;     /* 32 bytes per feature (mDescriptors)                       */
    const int32_t *counts;          /* features per frame (N)                                    */
    const int32_t *groups;          /* DBoW2 node id per feature (FeatureVector, src/Frame.cc:889-892);
                                       negative = the feature is not filed in the FeatureVector and
                                       is never matched; NULL = every feature in one node = brute force */
    const uint8_t *valid;           /* 1 = feature has a non-bad MapPoint (src/ORBmatcher.cc:268-274);
                                       NULL = all valid                                          */
    int capacity;
    int nframes;
} orbx_feature_set;

typedef struct orbx_bow_params {
    float nn_ratio;        /* mfNNratio  (ORBmatcher.h:57; 0.7 at src/Tracking.cc:1189)          */
    int check_orientation; /* mbCheckOrientation                                                 */
    int mode;              /* 0: SearchByBoW(KeyFrame*,Frame&,...)     src/ORBmatcher.cc:230-382,
                                 result indexed by the Frame feature (value = KeyFrame feature)
                              1: SearchByBoW(KeyFrame*,KeyFrame*,...)  src/ORBmatcher.cc:656-799,
                                 result indexed by the KF1 feature (value = KF2 feature)         */
} orbx_bow_params;

int orbx_matcher_create(int device, int max_features, int max_pairs, orbx_matcher **out);
void orbx_matcher_destroy(orbx_matcher *m);

/* SearchByBoW for `npairs` independent (A frame, B frame) pairs; A plays the KeyFrame.
 * a/b hold DEVICE pointers; pairs_a/pairs_b are host arrays of frame indices.  Asynchronous
 * on the matcher's stream (which first waits for `after_stream_of`, an extractor whose
 * outputs are being consumed; may be NULL).  Results stay on the device:
 *   matches[p*stride + slot] = index of the matched feature in the other set or -1,
 *   dists[p*stride + slot]   = Hamming distance of that match,
 *   nmatches[p]              = return value of SearchByBoW.
 * The Hamming distances are XOR + population count on the vector ALUs (DescriptorDistance, src/ORBmatcher.cc:1913-1933).  Environment
 * ORBX_MATCH_MFMA=1 (read per call) takes the candidate lists of the unfiltered case - no node ids, no validity mask - from the matrix
 * cores instead (v_mfma_i32_32x32x32_i8 on +-1 / 0-1 bytes: bit-identical results); a measured alternative, not the default.            */
int orbx_search_by_bow_device(orbx_matcher *m, const orbx_feature_set *a, const orbx_feature_set *b,
                              const int32_t *pairs_a, const int32_t *pairs_b, int npairs,
                              const orbx_bow_params *params, orbx_extractor *after_stream_of);

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo)
 * (src/ORBmatcher.cc:810-1017; LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:332).  The
 * feature sets are the two KeyFrames: keypoints = mvKeysUn, groups = mFeatVec node ids,
 * valid = 1 where the feature may be matched: it has NO MapPoint and, when bOnlyStereo, mvuRight >= 0
 * (:845-855, 867-876).  Acceptance per KF1 feature, in FeatureVector order: among the still unmatched
 * KF2 features of its node with dist <= TH_LOW that pass the epipole gate (:888-895, both monocular)
 * and CheckDistEpipolarLine (:188-227), the smallest distance, the LAST one among equals (:880);
 * then the rotation histogram.  matches[p*stride + i] = KF2 feature of KF1 feature i or -1
 * (vMatchedPairs = the non-negative entries in ascending i), nmatches[p] = return value. */
typedef struct orbx_triangulation_params {
    const float *f12;           /* HOST [9*npairs]: F12 row-major as LocalMapping::ComputeF12 returns it   */
    const float *epipole;       /* HOST [2*npairs]: ex, ey = KF1's centre projected into KF2 (:817-826)    */
    const uint8_t *stereo_a;    /* pKF1->mvuRight[i] >= 0, laid out like set a (device pointer in the
                                   _device form, host pointer in the host form); NULL = monocular          */
    const uint8_t *stereo_b;    /* same for pKF2                                                           */
    const float *scale_factors; /* HOST pKF2->mvScaleFactors[nlevels]                                      */
    const float *level_sigma2;  /* HOST pKF2->mvLevelSigma2[nlevels]                                       */
    int nlevels;
    int check_orientation;      /* mbCheckOrientation                                                      */
} orbx_triangulation_params;
int orbx_search_for_triangulation_device(orbx_matcher *m, const orbx_feature_set *a, const orbx_feature_set *b,
                                         const int32_t *pairs_a, const int32_t *pairs_b, int npairs,
                                         const orbx_triangulation_params *params, orbx_extractor *after_stream_of);
/* Host-array form for one KeyFrame pair (upload, run, download): matches12[a->counts[0]]. */
int orbx_search_for_triangulation(orbx_matcher *m, const orbx_feature_set *a_host, const orbx_feature_set *b_host,
                                  const orbx_triangulation_params *params_host, int32_t *matches12, int32_t *nmatches);

/* Hamming stage of Frame::ComputeStereoMatches (src/Frame.cc:1041-1216) for `npairs`
 * (left frame, right frame) pairs: per left keypoint the right keypoint of minimum
 * descriptor distance among those in its row band (+-2*scale[octave]), within one octave and
 * with uR in [uL - max_disparity, uL].  dists = bestDist (TH_HIGH=100 when none),
 * matches = bestIdxR (0 when none), nmatches[p] = #left keypoints with bestDist < 75.       */
int orbx_stereo_match_device(orbx_matcher *m, const orbx_feature_set *left, const orbx_feature_set *right,
                             const int32_t *pairs_l, const int32_t *pairs_r, int npairs,
                             const float *scale_factors, int nlevels, float max_disparity,
                             orbx_extractor *after_stream_of);

/* Complete Frame::ComputeStereoMatches (src/Frame.cc:1026-1420) for `npairs` (left frame, right
 * frame) pairs taken from the LAST batches of two extractor handles (the reference's
 * mpORBextractorLeft / mpORBextractorRight; both may be the same handle when left and right
 * frames were extracted in one batch): Hamming stage as orbx_stereo_match_device, then the 11x11
 * SAD search over +-5 px on the keypoint's pyramid level (the extractor's device-resident
 * mvImagePyramid), parabola sub-pixel fit, disparity gate and the median*1.5*1.4 outlier cut.
 *   uright[p*stride + iL] = mvuRight[iL] (-1: no match), depth[...] = mvDepth[iL],
 *   nmatches[p] = number of left keypoints with a depth; matches/dists = bestIdxR/bestDist.
 * mbf = Frame::mbf, mb = Frame::mb AS IT IS when ComputeStereoMatches runs: 0 in this fork's
 * stereo constructor (src/Frame.cc:125, mb is assigned at :197 after the call) => maxD = +inf.
 * Asynchronous on the matcher's stream, ordered after both extractors; the extractors' next
 * batch waits for it before their pyramids are overwritten. */
int orbx_compute_stereo_matches_device(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right,
                                       const int32_t *frames_l, const int32_t *frames_r, int npairs, float mbf,
                                       float mb);
int orbx_stereo_results_device(orbx_matcher *m, const float **uright_dev, const float **depth_dev, int *stride);
int orbx_stereo_download(orbx_matcher *m, int npairs, float *uright, float *depth, int stride);
/* Frame::ComputeStereoMatches (src/Frame.cc:1026-1420) for ONE stereo frame whose left / right image were extracted by the two
 * extractors' last single-frame calls (orbx_extract_view* / orbx_extract: the stereo Frame constructor, src/Frame.cc:159-168):
 * uright[i] / depth[i] = mvuRight / mvDepth of left keypoint i, i < n.  Synchronous; the latency form of
 * orbx_compute_stereo_matches_device + orbx_stereo_download (same kernels, three launches and one wait instead of ~22 runtime calls). */
int orbx_stereo_frame(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, float mbf, float mb, float *uright, float *depth, int n);
/* The same call in two halves: _begin launches and returns at once, _end waits and copies mvuRight / mvDepth out.  Neither extractor may
 * be called in between.  shim/Frame_hip.cc begins on the extractor thread that finishes last in the stereo constructor (src/Frame.cc:
 * 159-167), before that thread converts its keypoints, and ends in Frame::ComputeStereoMatches (:168): the match runs while the host
 * fills mvKeysRight and joins the threads. */
int orbx_stereo_frame_begin(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, float mbf, float mb);
int orbx_stereo_frame_end(orbx_matcher *m, float *uright, float *depth, int n);

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)
 * (ORBmatcher.h, src/ORBmatcher.cc:70-175; called by Tracking::SearchLocalPoints, src/Tracking.cc:1616)
 * including Frame::GetFeaturesInArea and the 64x48 feature grid (src/Frame.cc:741-877).
 * The Frame side is passed as arrays of the frame's members, the MapPoint side as what
 * Frame::isInFrustum left in every MapPoint.  Feature i of frame f at f*capacity + i. */
typedef struct orbx_projection_frame {
    const orbx_keypoint *keypoints_un; /* mvKeysUn                                                   */
    const uint8_t *descriptors;        /* mDescriptors                                               */
    const float *u_right;              /* mvuRight (<= 0: no stereo coordinate)                      */
    const uint8_t *occupied;           /* 1 = mvpMapPoints[i] holds a MapPoint with Observations()>0
                                          (src/ORBmatcher.cc:110-112); NULL = none                   */
    const int32_t *counts;             /* N per frame                                                */
    int capacity, nframes;
    float min_x, min_y;                /* Frame::mnMinX, mnMinY                                      */
    float grid_width_inv, grid_height_inv; /* Frame::mfGridElementWidthInv / HeightInv              */
} orbx_projection_frame;

typedef struct orbx_projection_points {
    const float *proj_x, *proj_y, *proj_xr; /* MapPoint::mTrackProjX / mTrackProjY / mTrackProjXR    */
    const int32_t *scale_level;        /* mnTrackScaleLevel                                          */
    const float *view_cos;             /* mTrackViewCos                                              */
    const uint8_t *in_view;            /* mbTrackInView && !isBad()                                  */
    const uint8_t *has_observations;   /* Observations()>0: the point blocks its feature for the
                                          points that follow; NULL = all                             */
    const uint8_t *descriptors;        /* GetDescriptor(), 32 bytes                                  */
    const int32_t *counts;             /* points per frame                                           */
    int capacity;
} orbx_projection_points;

/* Device-pointer form for `frame->nframes` independent (frame, point list) problems (the replay keeps 6 bytes of LDS
 * per feature slot: frame->capacity <= 27000, ORBX_ERR_CAPACITY beyond); results:
 * matches[f*stride + i] = index of the point written into F.mvpMapPoints[i] by the call or -1,
 * nmatches[f] = return value (orbx_matcher_results_device / orbx_matcher_download). */
int orbx_search_by_projection_device(orbx_matcher *m, const orbx_projection_frame *frame,
                                     const orbx_projection_points *points, const float *scale_factors, int nlevels,
                                     float th, float nn_ratio);
/* Host-array form for one frame (counts[0] entries each): upload, run, download. */
int orbx_search_by_projection(orbx_matcher *m, const orbx_projection_frame *frame_host,
                              const orbx_projection_points *points_host, const float *scale_factors, int nlevels,
                              float th, float nn_ratio, int32_t *assigned, int32_t *nmatches);

/* Frame::isInFrustum(pMP, viewingCosLimit) (reference src/Frame.cc:608-742) for a list of map points per frame: the
 * loop of Tracking::SearchLocalPoints (src/Tracking.cc:1580-1613) that prepares SearchByProjection(F, vpMapPoints, th).
 * Outputs are laid out like orbx_projection_points (point i of frame f at f*capacity + i), so that
 * orbx_frustum_results_device feeds orbx_search_by_projection_device without a host round trip:
 * in_view = mbTrackInView, proj_x / proj_y / proj_xr = mTrackProjX / Y / XR, scale_level = mnTrackScaleLevel,
 * view_cos = mTrackViewCos (only written where in_view). */
typedef struct orbx_frustum_frame {
    const float *tcw;                 /* [16] per frame: mTcw row-major (mRcw, mtcw, mOw follow as in Frame::UpdatePoseMatrices) */
    float fx, fy, cx, cy, mbf;        /* Frame statics                                                             */
    float min_x, max_x, min_y, max_y; /* mnMinX .. mnMaxY                                                          */
    const float *ratio_thresholds;    /* HOST [nlevels-1] from orbx_predict_scale_thresholds(mfLogScaleFactor, ..) */
    int nlevels;                      /* mnScaleLevels                                                             */
    int nframes;
} orbx_frustum_frame;
typedef struct orbx_map_points {
    const float *world_pos;     /* [3] GetWorldPos()                                                                */
    const float *normal;        /* [3] GetNormal()                                                                  */
    const float *max_distance;  /* mfMaxDistance (GetMaxDistanceInvariance()/1.2f)                                  */
    const float *min_distance;  /* mfMinDistance                                                                    */
    const int32_t *counts;      /* points per frame                                                                 */
    int capacity;
} orbx_map_points;
/* MapPoint::PredictScale (src/MapPoint.cc:571-586) without a device logarithm: thresholds[k] = the largest float ratio
 * mfMaxDistance/dist that the reference's ceil(log(ratio)/mfLogScaleFactor) still maps to level <= k (k = 0..nlevels-2),
 * tabulated on the host with the libm log the reference itself calls. */
int orbx_predict_scale_thresholds(float log_scale_factor, int nlevels, float *thresholds);
int orbx_is_in_frustum_device(orbx_matcher *m, const orbx_frustum_frame *frame, const orbx_map_points *points,
                              float viewing_cos_limit);
int orbx_frustum_results_device(orbx_matcher *m, const float **proj_x, const float **proj_y, const float **proj_xr,
                                const int32_t **scale_level, const float **view_cos, const uint8_t **in_view);
/* Host-array form for one frame (points->counts[0] points). */
int orbx_is_in_frustum(orbx_matcher *m, const orbx_frustum_frame *frame_host, const orbx_map_points *points_host,
                       float viewing_cos_limit, float *proj_x, float *proj_y, float *proj_xr, int32_t *scale_level,
                       float *view_cos, uint8_t *in_view);

/* Tracking::SearchLocalPoints (src/Tracking.cc:1760-1830) as ONE device chain for one frame: Frame::isInFrustum over the local
 * map points (src/Frame.cc:608-742, the loop at src/Tracking.cc:1791-1811) feeding ORBmatcher::SearchByProjection(Frame&,
 * vector<MapPoint*>&, th) (src/ORBmatcher.cc:70-175; :1828) without the mTrack* fields ever visiting the host: one upload of the
 * frame side, the pose and the points' map data, k_is_in_frustum -> k_proj_topk -> k_proj_greedy, one read-back.
 * `points_host` lists the points the reference's loop would test (not yet seen in this frame, not bad), in list order.
 * Outputs: assigned[i] (frame->counts[0] entries) = index of the point written into F.mvpMapPoints[i] or -1, *nmatches = the
 * search's return value; per point the fields Frame::isInFrustum leaves in the MapPoint: in_view = mbTrackInView (= the function's
 * return value: IncreaseVisible() / nToMatch bookkeeping is the caller's), proj_x / proj_y / proj_xr / scale_level / view_cos
 * (written where in_view; any of the five may be NULL). */
typedef struct orbx_local_points {
    const float *world_pos;          /* [3] GetWorldPos()                                   */
    const float *normal;             /* [3] GetNormal()                                     */
    const float *max_distance;       /* mfMaxDistance                                       */
    const float *min_distance;       /* mfMinDistance                                       */
    const uint8_t *descriptors;      /* GetDescriptor(), 32 bytes                           */
    const uint8_t *has_observations; /* Observations()>0; NULL = all                        */
    int count;
} orbx_local_points;
int orbx_search_local_points(orbx_matcher *m, const orbx_projection_frame *frame_host, const orbx_frustum_frame *pose_host,
                             const orbx_local_points *points_host, const float *scale_factors, int nlevels,
                             float viewing_cos_limit, float th, float nn_ratio, int32_t *assigned, int32_t *nmatches,
                             uint8_t *in_view, float *proj_x, float *proj_y, float *proj_xr, int32_t *scale_level,
                             float *view_cos);

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th,
 * const bool bMono) (src/ORBmatcher.cc:1569-1728; Tracking::TrackWithMotionModel,
 * src/Tracking.cc:1433-1441).  Last-frame side, feature i of frame f at f*capacity + i: */
typedef struct orbx_projection_last {
    const uint8_t *valid;            /* 1 = LastFrame.mvpMapPoints[i] != NULL && !mvbOutlier[i]             */
    const float *world_pos;          /* [3] MapPoint::GetWorldPos()                                         */
    const uint8_t *descriptors;      /* MapPoint::GetDescriptor(), 32 bytes                                 */
    const uint8_t *has_observations; /* Observations()>0; NULL = all                                        */
    const int32_t *octave;           /* LastFrame.mvKeys[i].octave                                          */
    const float *angle;              /* LastFrame.mvKeysUn[i].angle                                         */
    const int32_t *counts;           /* LastFrame.N per frame                                               */
    int capacity;
    const float *tcw_current;        /* [16] per frame: CurrentFrame.mTcw, row-major                        */
    const float *tcw_last;           /* [16] per frame: LastFrame.mTcw                                      */
    float fx, fy, cx, cy, mbf, mb;   /* CurrentFrame.fx .. mb                                               */
    float max_x, max_y;              /* Frame::mnMaxX, mnMaxY (min_x/min_y are in orbx_projection_frame)    */
} orbx_projection_last;
/* matches[f*stride + i2] = last-frame feature whose MapPoint ends up in CurrentFrame.mvpMapPoints[i2],
 * -1 = the call did not touch the feature, -2 = assigned and then cleared by the rotation-histogram
 * pruning (the reference stores NULL there, :1718); nmatches[f] = return value. */
int orbx_search_by_projection_last_device(orbx_matcher *m, const orbx_projection_frame *frame,
                                          const orbx_projection_last *last, const float *scale_factors, int nlevels,
                                          float th, int b_mono, int check_orientation);
int orbx_search_by_projection_last(orbx_matcher *m, const orbx_projection_frame *frame_host,
                                   const orbx_projection_last *last_host, const float *scale_factors, int nlevels,
                                   float th, int b_mono, int check_orientation, int32_t *assigned, int32_t *nmatches);

/* ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:1020-1177; LocalMapping::SearchInNeighbors)
 * and ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:1179-1312; LoopClosing): the search of
 * steps 2-3 for every map point - KeyFrame::GetFeaturesInArea(u, v, radius), the level gate, for the
 * first overload the chi-square gate on the reprojection error (:1111-1135, chi2_gate = 1) and the
 * feature of minimum Hamming distance (first minimum in GetFeaturesInArea order).  What couples the
 * map points in the reference's loop (isBad / IsInKeyFrame, Replace, AddObservation, :1036-1040,
 * 1150-1172) is sequential pointer surgery and stays with the caller, evaluated in order on the result.
 * kf: mvKeysUn, mDescriptors, mvuRight and the grid statics of the KeyFrame (`occupied` is not read). */
typedef struct orbx_fuse_points {
    const float *u, *v;          /* projection of the map point into the KeyFrame (:1056-1060)          */
    const float *ur;             /* u - bf*invz (:1066); read only with chi2_gate                       */
    const int32_t *level;        /* nPredictedLevel = pMP->PredictScale(dist3D, pKF) (:1090)            */
    const float *radius;         /* th*pKF->mvScaleFactors[nPredictedLevel] (:1093)                     */
    const uint8_t *active;       /* 1 = the point reached step 2 (gates :1036-1088); NULL = all         */
    const uint8_t *descriptors;  /* pMP->GetDescriptor(), 32 bytes                                      */
    const int32_t *counts;       /* points per KeyFrame                                                 */
    int capacity;
    float kf_min_x, kf_min_y;    /* (float)pKF->mnMinX / mnMinY: KeyFrame stores the bounds as int
                                    (include/KeyFrame.h), and its GetFeaturesInArea computes the cell window
                                    with them, while mGrid was filed with the Frame's float bounds
                                    (kf->min_x / min_y)                                                  */
} orbx_fuse_points;
/* Device-pointer form for kf->nframes independent (KeyFrame, point list) problems; results:
 * matches[f*stride + i] = bestIdx of point i or -1, dists[f*stride + i] = bestDist (256 when none)
 * (orbx_matcher_results_device / orbx_matcher_download); the caller fuses when bestDist <= TH_LOW (:1148).
 * inv_level_sigma2: HOST pKF->mvInvLevelSigma2[nlevels]. */
int orbx_fuse_search_device(orbx_matcher *m, const orbx_projection_frame *kf, const orbx_fuse_points *points,
                            const float *inv_level_sigma2, int nlevels, int chi2_gate);
/* Host-array form for one KeyFrame (counts[0] entries each): best_idx / best_dist per map point. */
int orbx_fuse_search(orbx_matcher *m, const orbx_projection_frame *kf_host, const orbx_fuse_points *points_host,
                     const float *inv_level_sigma2, int nlevels, int chi2_gate, int32_t *best_idx, int32_t *best_dist);

/* Greedy area search: the loops of ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th)
 * (loop closing, src/ORBmatcher.cc:388-513) and SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th,
 * ORBdist) (relocalisation, :1731-1864) after their per-point preparation (:410-452 / :1755-1790, cv::Mat
 * expressions and PredictScale, kept on the host by the shim).  Queries are processed in order; query i
 * takes the feature of minimum Hamming distance (first minimum in GetFeaturesInArea order) among the
 * features inside GetFeaturesInArea(u, v, radius[, min_level, max_level]) that are not blocked - blocked
 * from the start (frame->occupied: vpMatched[idx] / CurrentFrame.mvpMapPoints[i2] non-NULL) or taken by an
 * earlier query - provided that distance is <= max_dist (TH_LOW / ORBdist, < 256); the feature is then blocked.
 * Level gate: octave < min_level or (max_level >= 0 and octave > max_level) rejects (Frame::GetFeaturesInArea,
 * src/Frame.cc:741-850; the KeyFrame variants pass [level-1, level], :469-470). */
typedef struct orbx_area_queries {
    const float *u, *v, *radius;
    const int32_t *min_level, *max_level;
    const uint8_t *active;       /* 1 = the point reached the search; NULL = all                            */
    const uint8_t *descriptors;  /* pMP->GetDescriptor(), 32 bytes                                          */
    const int32_t *counts;       /* queries per frame                                                       */
    int capacity;
    float window_min_x, window_min_y; /* bounds used for the cell window: Frame::mnMinX/Y, or (float) of the
                                         KeyFrame's int mnMinX/Y (see orbx_fuse_points)                     */
} orbx_area_queries;
/* results: matches[f*stride + i] = feature taken by query i or -1, dists[...] its distance (256 when none),
 * nmatches[f] = number of queries that took a feature. */
int orbx_area_search_greedy_device(orbx_matcher *m, const orbx_projection_frame *frame, const orbx_area_queries *queries, int max_dist);
int orbx_area_search_greedy(orbx_matcher *m, const orbx_projection_frame *frame_host, const orbx_area_queries *queries_host, int max_dist,
                            int32_t *assigned, int32_t *dists, int32_t *nmatches);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)
 * (src/ORBmatcher.cc:515-654; Tracking::MonocularInitialization, src/Tracking.cc:944).  f1: mvKeysUn and
 * descriptors of the reference frame; f2: the current frame with its grid statics; prev_matched_xy: vbPrevMatched
 * (x, y per F1 feature, laid out like f1).  matches[f*stride + i1] = vnMatches12[i1], nmatches[f] = return value;
 * the caller refreshes vbPrevMatched from the matches (:646-650). */
int orbx_search_for_initialization_device(orbx_matcher *m, const orbx_feature_set *f1, const orbx_projection_frame *f2,
                                          const float *prev_matched_xy, int window_size, float nn_ratio, int check_orientation);
int orbx_search_for_initialization(orbx_matcher *m, const orbx_feature_set *f1_host, const orbx_projection_frame *f2_host,
                                   const float *prev_matched_xy, int window_size, float nn_ratio, int check_orientation,
                                   int32_t *matches12, int32_t *nmatches);

int orbx_matcher_results_device(orbx_matcher *m, const int32_t **matches_dev, const int32_t **dists_dev,
                                const int32_t **nmatches_dev, int *stride);
int orbx_matcher_download(orbx_matcher *m, int npairs, int32_t *matches, int32_t *dists, int stride,
                          int32_t *nmatches);
int orbx_matcher_sync(orbx_matcher *m);

/* Host-array convenience forms for one pair (upload, run, download). */
int orbx_search_by_bow(orbx_matcher *m, const orbx_feature_set *a_host, const orbx_feature_set *b_host,
                       const orbx_bow_params *params, int32_t *matches, int32_t *nmatches);
int orbx_stereo_match(orbx_matcher *m, const orbx_feature_set *left_host, const orbx_feature_set *right_host,
                      const float *scale_factors, int nlevels, float max_disparity, int32_t *best_dist,
                      int32_t *best_idx);
/* Average kernel milliseconds (HIP events on the matcher's stream) per *_device call since
 * the previous orbx_matcher_last_timing (at most the last 64 calls). */
int orbx_matcher_last_timing(orbx_matcher *m, float *total_ms);
/* Split of the SearchByBoW calls averaged by the previous orbx_matcher_last_timing: the distance /
 * candidate-list kernel (k_bow_topk) and the greedy replay with its preparation (k_bow_order + k_bow_greedy). */
int orbx_matcher_last_kernel_timing(orbx_matcher *m, float *distance_ms, float *replay_ms);


/* ------------------------------------------------------------------------------------
 * Bag of words  ==  DBoW2::TemplatedVocabulary<FORB::TDescriptor,FORB>::transform, the work of
 * Frame::ComputeBoW / KeyFrame::ComputeBoW (reference src/Frame.cc:880-896,
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1262): per feature the word id, the word
 * weight and the node `levelsup` levels above the leaf - the FeatureVector key that gates
 * ORBmatcher::SearchByBoW, directly usable as orbx_feature_set.groups.
 * ---------------------------------------------------------------------------------- */
typedef struct orbx_vocabulary orbx_vocabulary;

/* The tree as flat arrays, as ORBVocabulary::loadFromTextFile builds it
 * (TemplatedVocabulary.h:1338-1420): node 0 = root, parent[i] < i, the children of a node are
 * its child ids in ascending order, word ids count the leaves in node-id order.
 * descriptors: 32 bytes per node (root unused), weights: one double per node (leaves: word weight). */
int orbx_vocabulary_create(int device, int k, int L, int num_nodes, const int32_t *parent, const uint8_t *is_leaf,
                           const uint8_t *descriptors, const double *weights, orbx_vocabulary **out);
void orbx_vocabulary_destroy(orbx_vocabulary *v);
int orbx_vocabulary_words(const orbx_vocabulary *v);
/* transform() of every feature of the extractor's LAST batch, on the extractor's stream (ordered
 * behind the extraction and ahead of any consumer that orders itself behind the extractor).
 * Results stay on the device, laid out like the extractor's results (feature i of frame f at
 * f*capacity + i): word id, FeatureVector node id (-1 when the word's weight is 0: the reference
 * does not file such a feature, :1160-1166) and the word weight. */
int orbx_bow_transform_device(orbx_vocabulary *v, orbx_extractor *ext, int levelsup);
int orbx_bow_results_device(orbx_vocabulary *v, const int32_t **word_dev, const int32_t **node_dev,
                            const double **weight_dev, int *capacity);
int orbx_bow_download(orbx_vocabulary *v, orbx_extractor *ext, int batch, int32_t *word, int32_t *node, double *weight);
/* Host-array form for n descriptors (upload, run, download). */
int orbx_bow_transform(orbx_vocabulary *v, const uint8_t *descriptors, int n, int levelsup, int32_t *word,
                       int32_t *node, double *weight);
/* The same call, plus the two orders in which transform() fills its std::map results (TemplatedVocabulary.h:1146-1196): by_word[k] / by_node[k] =
 * the feature that is k-th in ascending (word id, feature index) / (node id, feature index) order among the *filed features (word weight > 0).  A caller
 * that builds the BowVector / FeatureVector from them inserts every key at the end of the map (emplace_hint) and accumulates / appends in feature order,
 * i.e. gets the reference's maps bit for bit without ~2000 tree searches (shim/BoW_hip.cc: 66 -> 20 us per frame).  The orders are ranked on the device. */
int orbx_bow_transform_sorted(orbx_vocabulary *v, const uint8_t *descriptors, int n, int levelsup, int32_t *word, int32_t *node, double *weight,
                              int32_t *by_word, int32_t *by_node, int32_t *filed);
/* Latency form for the ONE frame that `ext`'s last single-frame call extracted (Frame::ComputeBoW of the frame the constructor has just built,
 * src/Frame.cc:880-896 behind :394-456): the descriptors are read where the extractor left them on the device - nothing is uploaded - and the call
 * can be begun the moment the extraction is complete, long before the tracking thread asks for the BowVector.  A job owns its stream, result
 * buffers and pinned memory and only READS the vocabulary: any number of jobs and orbx_bow_transform* calls may use one vocabulary concurrently
 * (the vocabulary must outlive its jobs).  _begin returns at once; _end waits and hands out pointers INTO the job's pinned memory, valid until the
 * next call on the job: word / node / weight per feature, the two orders of orbx_bow_transform_sorted, *filed and the feature count *n.
 * `ext` must not be called between _begin and the completion of the job's kernels (~20 us). */
typedef struct orbx_bow_job orbx_bow_job;
int orbx_bow_job_create(orbx_vocabulary *v, orbx_bow_job **out);
void orbx_bow_job_destroy(orbx_bow_job *j);
int orbx_bow_job_begin(orbx_bow_job *j, orbx_extractor *ext, int levelsup);
int orbx_bow_job_end(orbx_bow_job *j, const int32_t **word, const int32_t **node, const double **weight, const int32_t **by_word, const int32_t **by_node,
                     int32_t *filed, int32_t *n);


/* ------------------------------------------------------------------------------------
 * Rest of the Frame constructor  ==  Frame::UndistortKeyPoints (reference src/Frame.cc:899-947),
 * Frame::ComputeImageBounds (:950-1004) and Frame::AssignFeaturesToGrid / PosInGrid
 * (:460-491, 868-878); called from both Frame constructors (src/Frame.cc:181, 210, 234 and
 * 422-456).  One handle per camera (mK, mDistCoef).
 * ---------------------------------------------------------------------------------- */
#define ORBX_FRAME_GRID_COLS 64 /* FRAME_GRID_COLS, include/Frame.h:55-60 */
#define ORBX_FRAME_GRID_ROWS 48 /* FRAME_GRID_ROWS                         */
typedef struct orbx_frame_ops orbx_frame_ops;
typedef struct orbx_camera {
    float fx, fy, cx, cy; /* mK                                                                   */
    float dist[5];        /* mDistCoef: k1 k2 p1 p2 [k3]; dist[0] == 0 means "already rectified",
                             the reference's own test (src/Frame.cc:901, 953)                     */
    int ndist;            /* 4 or 5 (src/Tracking.cc:127-136 appends k3 only when non-zero)       */
} orbx_camera;
/* The statics PosInGrid reads (src/Frame.cc:868-878): Frame::mnMinX, mnMinY,
 * mfGridElementWidthInv = 64/(mnMaxX-mnMinX), mfGridElementHeightInv = 48/(mnMaxY-mnMinY) (:326-327). */
typedef struct orbx_frame_grid {
    float min_x, min_y, width_inv, height_inv;
} orbx_frame_grid;

int orbx_frame_ops_create(int device, const orbx_camera *camera, orbx_frame_ops **out);
void orbx_frame_ops_destroy(orbx_frame_ops *h);
/* Frame::ComputeImageBounds(imLeft): bounds = mnMinX, mnMaxX, mnMinY, mnMaxY for a cols x rows image. */
int orbx_frame_image_bounds(orbx_frame_ops *h, int cols, int rows, float *bounds);
/* Frame::UndistortKeyPoints on n host keypoints (upload, run, download): kp_un[i] = keypoints[i]
 * with pt replaced by the undistorted point. */
int orbx_frame_undistort(orbx_frame_ops *h, const orbx_keypoint *keypoints, int n, orbx_keypoint *kp_un);
/* Frame::AssignFeaturesToGrid on n host mvKeysUn: mGrid as CSR, cell = x*ORBX_FRAME_GRID_ROWS + y,
 * grid_offsets[cell .. cell+1] (64*48+1 entries) delimit the feature indices of mGrid[x][y] inside
 * grid_indices[n], in the reference's push_back (= ascending) order. */
int orbx_frame_assign_grid(orbx_frame_ops *h, const orbx_frame_grid *grid, const orbx_keypoint *kp_un, int n,
                           int32_t *grid_offsets, int32_t *grid_indices);
/* Both, fused, for every frame of the extractor's LAST batch on the extractor's stream.  Results stay
 * on the device: mvKeysUn laid out like the extractor's keypoints (feature i of frame f at
 * f*capacity + i), grid_offsets[f*(64*48+1) + ...], grid_indices[f*capacity + ...]. */
int orbx_frame_finish_device(orbx_frame_ops *h, orbx_extractor *ext, const orbx_frame_grid *grid);
int orbx_frame_results_device(orbx_frame_ops *h, const orbx_keypoint **kp_un_dev, const int32_t **grid_offsets_dev,
                              const int32_t **grid_indices_dev, int *capacity);
int orbx_frame_download(orbx_frame_ops *h, orbx_extractor *ext, int batch, orbx_keypoint *kp_un, int32_t *grid_offsets,
                        int32_t *grid_indices);
/* Latency form for ONE frame that `ext`'s last single-frame call extracted (the Frame constructors, src/Frame.cc:168-234, 394-456):
 * _begin launches the fused kernel on the handle's own stream, reading the keypoints where the extractor left them on the device and
 * writing mvKeysUn / the grid into the handle's pinned memory, and returns at once (grid == NULL: undistort only); _end waits and hands
 * out pointers INTO that pinned memory, valid until the next call on the handle: kp_un (NULL when the camera is not distorted: mvKeysUn
 * = mvKeys, src/Frame.cc:901-905), grid_offsets[64*48+1], grid_indices[n].  `ext` must not be called in between. */
int orbx_frame_finish_begin(orbx_frame_ops *h, orbx_extractor *ext, const orbx_frame_grid *grid);
int orbx_frame_finish_end(orbx_frame_ops *h, const orbx_keypoint **kp_un, const int32_t **grid_offsets, const int32_t **grid_indices, int *n);


/* ------------------------------------------------------------------------------------
 * Local bundle adjustment  ==  the numerical core of Optimizer::LocalBundleAdjustment
 * (reference include/Optimizer.h:112, src/Optimizer.cc:629-997): g2o BlockSolver_6_3 +
 * Levenberg-Marquardt with Schur complement, 5 robust (Huber) iterations, outlier
 * re-classification, 10 non-robust iterations.  The class shim collects the local window
 * from the KeyFrame/MapPoint graph (src/Optimizer.cc:634-853) into these flat arrays and
 * writes poses/points/outliers back under the map mutex (:961-996).
 * Precision at the boundary is float32 like the reference's cv::Mat (src/Converter.cc);
 * everything inside is FP64.  Contract vs the CPU oracle: |delta| <= 1e-5.
 * ---------------------------------------------------------------------------------- */
typedef struct orbx_lba orbx_lba;

typedef struct orbx_lba_problem {
    int num_keyframes;            /* local + fixed keyframes                                      */
    const float *poses;           /* [K*16] Tcw, row-major 4x4 (KeyFrame::GetPose)                */
    const uint8_t *fixed;         /* [K] 1 = fixed vertex (lFixedCameras, or mnId==0)             */
    const float *intrinsics;      /* [K*5] fx, fy, cx, cy, mbf                                    */
    int num_points;
    const float *points;          /* [P*3] MapPoint::GetWorldPos                                  */
    int num_edges;                /* observations, in optimizer.addEdge order                     */
    const int32_t *edge_point;    /* [E]                                                          */
    const int32_t *edge_keyframe; /* [E]                                                          */
    const float *edge_obs;        /* [E*3] kpUn.pt.x, kpUn.pt.y, mvuRight (<0: monocular edge)    */
    const float *edge_inv_sigma2; /* [E] mvInvLevelSigma2[kpUn.octave]                            */
} orbx_lba_problem;

typedef struct orbx_lba_result {
    float *poses;           /* [K*16] optimised Tcw (fixed keyframes returned unchanged)          */
    float *points;          /* [P*3]                                                              */
    double *edge_chi2;      /* [E] e->chi2() as read at src/Optimizer.cc:921-958 (may be NULL)     */
    uint8_t *edge_outlier;  /* [E] 1 = goes to vToErase (chi2 > 5.991/7.815 or depth <= 0)          */
    double stats[8];        /* stage1 {iterations, LM trials, chi2 start, chi2 end}, stage2 idem    */
} orbx_lba_result;

int orbx_lba_create(int device, int max_keyframes, int max_points, int max_edges, orbx_lba **out);
void orbx_lba_destroy(orbx_lba *h);
/* stop_flag == pbStopFlag (mbAbortBA): polled before starting, between LM iterations and
 * between LM trials, like g2o's forceStopFlag; may be NULL. */
int orbx_lba_solve(orbx_lba *h, const orbx_lba_problem *problem, const volatile uint8_t *stop_flag,
                   orbx_lba_result *result);
/* Optimizer::BundleAdjustment / GlobalBundleAdjustemnt (reference src/Optimizer.cc:55-84, 86-360; Tracking::
 * CreateInitialMapMonocular with 20 iterations, LoopClosing::RunGlobalBundleAdjustment with 10): the same graph
 * and solver as the local window, one optimize(iterations) with Huber kernels iff robust, no outlier pass.
 * problem: every non-bad KeyFrame (fixed[k] = mnId == 0) and MapPoint with its observations; result as
 * orbx_lba_solve (edge_outlier / edge_chi2 = the final classification, informative only here).
 * The reduced (keyframe) system is DENSE here: its lower triangle is factored by the blocked Cholesky of csrc/orbx_lba.hip, up to
 * 24576 unknowns = 4096 free keyframes (CHOL_DENSE_MAX_N), ORBX_ERR_CAPACITY beyond.  Up to ~530 free keyframes a block row of
 * the Schur complement is accumulated in LDS; larger windows take the global-memory path (tested at 560 keyframes). */
int orbx_bundle_adjustment(orbx_lba *h, const orbx_lba_problem *problem, int iterations, int robust,
                           const volatile uint8_t *stop_flag, orbx_lba_result *result);
/* Kernel milliseconds (HIP events) spent inside the last orbx_lba_solve and FP64 flop count. */
int orbx_lba_last_timing(orbx_lba *h, float *device_ms, double *flops);


/* ------------------------------------------------------------------------------------
 * Motion-only bundle adjustment  ==  Optimizer::PoseOptimization(Frame *pFrame)
 * (reference include/Optimizer.h, src/Optimizer.cc:363-605; called by every Track* function of
 * Tracking): one SE3 vertex, unary EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose
 * edges with Huber kernels, 4 rounds of 10 Levenberg iterations each restarted from the initial
 * pose, inlier re-classification between rounds (chi2 > 5.991 / 7.815).  A batch of independent
 * frames is solved by one kernel launch (one workgroup per frame, no host round trip).
 * Feature i of frame f at f*capacity + i; only features that have a MapPoint are passed. */
typedef struct orbx_pose_optimizer orbx_pose_optimizer;
typedef struct orbx_pose_problem {
    int num_frames, capacity;
    const float *poses;        /* [B*16] pFrame->mTcw, row-major                                  */
    const float *cameras;      /* [B*5]  fx, fy, cx, cy, mbf                                      */
    const int32_t *counts;     /* [B]    features with a MapPoint (nInitialCorrespondences)        */
    const float *world_points; /* [B*cap*3] MapPoint::GetWorldPos()                                */
    const float *observations; /* [B*cap*3] kpUn.pt.x, kpUn.pt.y, mvuRight (< 0: monocular edge)   */
    const float *inv_sigma2;   /* [B*cap]   mvInvLevelSigma2[kpUn.octave]                          */
} orbx_pose_problem;
int orbx_pose_optimizer_create(int device, int max_frames, int max_features, orbx_pose_optimizer **out);
void orbx_pose_optimizer_destroy(orbx_pose_optimizer *h);
/* poses_out [B*16] = pFrame->mTcw after SetPose; outlier [B*cap] = pFrame->mvbOutlier of the passed
 * features; inliers [B] = the return value (nInitialCorrespondences - nBad); stats [B*8] = per round
 * {LM iterations, final robustified chi2}.  Any output may be NULL.  Contract vs the CPU oracle:
 * |delta pose| <= 1e-5, identical outlier flags. */
int orbx_pose_optimization(orbx_pose_optimizer *h, const orbx_pose_problem *problem, float *poses_out, uint8_t *outlier,
                           int32_t *inliers, double *stats);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */

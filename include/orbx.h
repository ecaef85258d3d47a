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

/* ------------------------------------------------------------------------------------
 * Synthetic input frames (host, integer-only, deterministic).  Not part of the
 * reference: datasets are absent, every benchmark/parity input comes from here.
 * ---------------------------------------------------------------------------------- */
#define ORBX_SYNTH_LOW_TEXTURE 1  /* few shapes, +-2 noise: hits the minThFAST fallback */
#define ORBX_SYNTH_STEREO_RIGHT 2 /* right view of the same scene, per-shape disparity   */
int orbx_synth_frame(uint64_t seed, int width, int height, int stride, int flags, uint8_t *dst);

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
int orbx_extractor_capacity(const orbx_extractor *h);

/* ORBextractor::operator() (ORBextractor.h:110, src/ORBextractor.cc:1544-1668) for one
 * host image.  `keypoints` / `descriptors` hold `capacity` entries / capacity*32 bytes;
 * *count receives the real number (<= orbx_extractor_capacity()).  An empty image
 * (NULL / zero size) returns ORBX_OK with *count = 0 (reference: silent return). */
int orbx_extract(orbx_extractor *h, const uint8_t *image, int width, int height, int stride,
                 orbx_keypoint *keypoints, uint8_t *descriptors, int capacity, int *count);

/* Batched form: `batch` independent frames of identical size.  Frame f writes
 * keypoints[f*capacity ...], descriptors[f*capacity*32 ...], counts[f].
 * Equivalent to `batch` operator() calls; the frames are data-parallel on the GPU. */
int orbx_extract_batch(orbx_extractor *h, const uint8_t *const *images, int batch, int width,
                       int height, int stride, orbx_keypoint *keypoints, uint8_t *descriptors,
                       int capacity, int *counts);

/* Device-resident batch: images_dev points to DEVICE memory, frame f at
 * images_dev + f*frame_pitch (rows `stride` bytes apart).  Runs asynchronously on the
 * handle's stream; results stay in handle-owned device buffers until downloaded. */
int orbx_extract_batch_device(orbx_extractor *h, const void *images_dev, int batch, int width,
                              int height, int stride, size_t frame_pitch);
/* Device pointers of the last batch's results: keypoints[f*cap + i], descriptors
 * [(f*cap + i)*32], counts[f]; *capacity = per-frame capacity of those arrays. */
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

/* std::vector<cv::Mat> mvImagePyramid (ORBextractor.h:161, read by
 * Frame::ComputeStereoMatches, src/Frame.cc:1044,1248,1272,1281): size and bytes of
 * pyramid level `level` of frame `frame` of the last call.  blurred=1 returns the
 * Gaussian-blurred copy the descriptors were sampled from (src/ORBextractor.cc:1626-1634). */
int orbx_pyramid_level_size(const orbx_extractor *h, int width, int height, int level, int *w, int *hgt);
int orbx_download_pyramid(orbx_extractor *h, int frame, int level, int blurred, uint8_t *dst, int dst_stride);

/* Stage taps for the parity tests: FAST score map of a level
 * (0 = not a corner at minThFAST), the per-level candidate list in vToDistributeKeys
 * order (src/ORBextractor.cc:1089-1157) packed as x | y<<12 | score<<24 relative to the
 * border window origin, and the per-level keypoints after DistributeOctTree + IC_Angle
 * in level coordinates (src/ORBextractor.cc:1167-1198). */
int orbx_debug_download_scores(orbx_extractor *h, int frame, int level, uint8_t *dst, int dst_stride);
int orbx_debug_download_candidates(orbx_extractor *h, int frame, int level, uint32_t *packed, int cap, int *count);
int orbx_debug_download_level_keypoints(orbx_extractor *h, int frame, int level, orbx_keypoint *kps, int cap, int *count);

/* Timing of the last orbx_extract_batch_device call, measured with HIP events on the
 * handle's stream: total milliseconds, and per-stage milliseconds in the order given
 * by orbx_stage_name(i).  Only filled when profiling was enabled before the call. */
#define ORBX_MAX_STAGES 16
int orbx_extractor_set_profiling(orbx_extractor *h, int enable);
int orbx_extractor_last_timing(orbx_extractor *h, float *total_ms, float *stage_ms, int *nstages);
const char *orbx_stage_name(int stage);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */

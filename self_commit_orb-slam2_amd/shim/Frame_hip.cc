// shim/Frame_hip.cc -- HIP bodies for ORB_SLAM2::Frame::ComputeStereoMatches, UndistortKeyPoints,
// ComputeImageBounds and AssignFeaturesToGrid (+ Frame::ExtractORB, which starts them early).
//
// Compiled against the REFERENCE's own include/Frame.h with shim/ORBextractor.h in place of
// include/ORBextractor.h.  Replaces the body of
//     void Frame::ComputeStereoMatches()                                      src/Frame.cc:1026-1420
// (called from the stereo constructor, src/Frame.cc:168).  The two extractors already hold the
// keypoints, descriptors and the unblurred pyramids of the left / right image on the device
// (the reference reads mpORBextractorLeft->mvImagePyramid, :1044, 1248, 1272, 1281), so nothing
// is uploaded: one call, then mvuRight / mvDepth come back.
// Note mb: the reference's stereo constructor initialises mb(0) (:125) and only assigns
// mb = mbf/fx AFTER this function (:197), so minZ = 0 and maxD = +inf; the member is passed
// through as is.
#include <stdexcept>
#include <string>
#include <vector>

#include "Frame.h"
#include "orbx.h"
#include "shim_error.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

// tells shim/ORBextractor.cc that ComputeStereoMatches reads the pyramid on the device: no host copy of mvImagePyramid per frame
extern "C" __attribute__((visibility("default"))) int orbx_shim_device_stereo_linked = 1;

// Where a Frame constructor spends its time (tools/latency_shim.py): accumulated wall time of every replaced member function.
#include <chrono>
enum { P_EXTRACT = 0, P_UNDISTORT, P_STEREO, P_BOUNDS, P_GRID, P_COUNT };
static double gProfUs[P_COUNT];
static unsigned long gProfCalls[P_COUNT];
static std::mutex gProfMutex;
namespace {
struct ShimTimer {
    int idx;
    std::chrono::steady_clock::time_point t0;
    explicit ShimTimer(int i) : idx(i), t0(std::chrono::steady_clock::now()) {}
    ~ShimTimer()
    {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> lock(gProfMutex);
        gProfUs[idx] += us; gProfCalls[idx]++;
    }
};
}  // namespace
// idx: 0 ExtractORB (per call: two per stereo frame, concurrent), 1 UndistortKeyPoints, 2 ComputeStereoMatches, 3 ComputeImageBounds, 4 AssignFeaturesToGrid;
// reset != 0 clears the counters after reading
extern "C" __attribute__((visibility("default"))) int orbx_shim_profile(int idx, int reset, double *total_us, unsigned long *calls)
{
    if (idx < 0 || idx >= P_COUNT) return -1;
    std::lock_guard<std::mutex> lock(gProfMutex);
    if (total_us) *total_us = gProfUs[idx];
    if (calls) *calls = gProfCalls[idx];
    if (reset) { gProfUs[idx] = 0; gProfCalls[idx] = 0; }
    return 0;
}

// Timeline of a constructor (tools/latency_shim.py --trace): orbx_shim_trace(1) starts recording (name, thread, microseconds) marks,
// orbx_shim_trace_dump prints them relative to the first one.  Off: one relaxed load per mark.
static std::atomic<int> gTraceOn(0), gTraceN(0);
struct TraceEv { const char *name; double us; unsigned long tid; };
static TraceEv gTrace[8192];
extern "C" __attribute__((visibility("default"))) void orbx_shim_trace_mark(const char *name)
{
    if (!gTraceOn.load(std::memory_order_relaxed)) return;
    const int i = gTraceN.fetch_add(1);
    if (i >= 8192) return;
    gTrace[i].name = name;
    gTrace[i].us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    gTrace[i].tid = (unsigned long)std::hash<std::thread::id>()(std::this_thread::get_id()) % 1000;
}
extern "C" __attribute__((visibility("default"))) void orbx_shim_trace(int on) { gTraceN.store(0); gTraceOn.store(on); }
extern "C" __attribute__((visibility("default"))) int orbx_shim_trace_dump(char *buf, int cap)
{
    int n = gTraceN.load(), o = 0;
    if (n > 8192) n = 8192;
    for (int i = 0; i < n && o < cap - 96; i++)
        o += snprintf(buf + o, (size_t)(cap - o), "%9.1f us  [thread %03lu]  %s\n", gTrace[i].us - gTrace[0].us, gTrace[i].tid, gTrace[i].name);
    if (cap > 0) buf[o < cap ? o : cap - 1] = 0;
    return n;
}
#define TRACE(name) orbx_shim_trace_mark(name)

// how often the early start was what produced a frame's mvKeysUn + mGrid / its stereo match (the tests check that the path they mean to test ran)
static unsigned long gEarlyFills = 0, gEarlyStereo = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_early_fills(void) { return gEarlyFills; }
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_early_stereo(void) { return gEarlyStereo; }
static unsigned long gStereoCalls = 0, gUndistortCalls = 0, gBoundsCalls = 0, gGridCalls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_undistort_calls(void) { return gUndistortCalls; }
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_image_bounds_calls(void) { return gBoundsCalls; }
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_assign_grid_calls(void) { return gGridCalls; }
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_compute_stereo_matches_calls(void) { return gStereoCalls; }

// shim/BoW_hip.cc (when it is part of the build): the device copy of an ORBVocabulary
extern "C" __attribute__((weak)) orbx_vocabulary *orbx_shim_device_vocabulary(const void *voc, unsigned long long *generation);

namespace ORB_SLAM2
{

namespace
{
// cv::KeyPoint <-> orbx_keypoint
void Pack(const std::vector<cv::KeyPoint> &in, std::vector<orbx_keypoint> &out)
{
    out.resize(in.size());
    for (size_t i = 0; i < in.size(); i++) {
        const cv::KeyPoint &k = in[i];
        orbx_keypoint &o = out[i];
        o.x = k.pt.x; o.y = k.pt.y; o.size = k.size; o.angle = k.angle; o.response = k.response; o.octave = k.octave; o.class_id = k.class_id;
    }
}

struct CamKey {
    float v[10];
    bool operator<(const CamKey &o) const
    {
        for (int i = 0; i < 10; i++) if (v[i] != o.v[i]) return v[i] < o.v[i];
        return false;
    }
    bool operator==(const CamKey &o) const { return memcmp(v, o.v, sizeof(v)) == 0; }
};

bool CameraOf(const cv::Mat &K, const cv::Mat &D, orbx_camera &cam, CamKey &key)
{
    cam.fx = K.at<float>(0, 0); cam.fy = K.at<float>(1, 1); cam.cx = K.at<float>(0, 2); cam.cy = K.at<float>(1, 2);
    cam.ndist = D.rows * D.cols;
    if (cam.ndist != 4 && cam.ndist != 5) return false;
    for (int i = 0; i < 5; i++) cam.dist[i] = i < cam.ndist ? D.at<float>(i) : 0.0f;
    const CamKey k = {{cam.fx, cam.fy, cam.cx, cam.cy, cam.dist[0], cam.dist[1], cam.dist[2], cam.dist[3], cam.dist[4], (float)cam.ndist}};
    key = k;
    return true;
}

// One device handle per camera (mK, mDistCoef), shared by the threads that build frames: the host-array forms.
std::mutex gOpsMutex;
std::map<CamKey, orbx_frame_ops *> gOps;
orbx_frame_ops *FrameOpsFor(const cv::Mat &K, const cv::Mat &D)
{
    orbx_camera cam;
    CamKey key;
    if (!CameraOf(K, D, cam, key)) { orbx_shim::Fail("Frame", "mDistCoef must hold 4 or 5 coefficients"); return 0; }
    std::lock_guard<std::mutex> lock(gOpsMutex);
    std::map<CamKey, orbx_frame_ops *>::iterator it = gOps.find(key);
    if (it != gOps.end()) return it->second;
    orbx_frame_ops *h = 0;
    if (orbx_frame_ops_create(orbx_shim::Device(), &cam, &h) != ORBX_OK) { orbx_shim::Fail("Frame"); return 0; }      // (not cached: the next frame tries again)
    gOps[key] = h;
    return h;
}
std::mutex gCallMutex;   // a handle is not re-entrant (include/orbx.h); frames are built by one thread at a time in the reference

// ---------------------------------------------------------------------------------------------
// What the constructors do after the extraction - UndistortKeyPoints, ComputeStereoMatches, AssignFeaturesToGrid (src/Frame.cc:181-234,
// 422-456) - needs only the extractors' results, and those are on the device the moment the extraction call's wait returns, while the
// host still has to convert keypoints for the caller, join its threads and walk through the constructor.  So the replaced
// Frame::ExtractORB STARTS that work from inside the extractor call (ORBextractor::SetPostExtract: after the device is done, before the
// conversion), and the member functions the constructor calls later only collect it:
//   left image done  -> undistortion + grid of the left keypoints launched (orbx_frame_finish_begin); collected at the end of the same
//                       ExtractORB call - the left thread finishes 30-40 us before the right one, which was started after it -, where
//                       mvKeysUn and mGrid are filled;  UndistortKeyPoints() / AssignFeaturesToGrid() find them done and return;
//   both images done -> the stereo match launched by whichever thread is second (orbx_stereo_frame_begin);  ComputeStereoMatches()
//                       waits for it and copies mvuRight / mvDepth.
// Every member function keeps its complete body for a caller that reaches it any other way (first frame: the grid statics do not exist
// yet, src/Frame.cc:203-221; a failed launch; ORBX_SHIM_EARLY=0).  State per LEFT extractor (a Frame is built by one constructor at a time
// per extractor: the handle is not re-entrant), hung on the shim's ORBextractor object.
// ---------------------------------------------------------------------------------------------
struct FrameAssist {
    std::atomic<int> arrivals, pairFailed;
    orbx_matcher *stereo;
    int stereoCap;
    const Frame *stereoFor;          // the frame whose match is in flight (begun, not ended)
    orbx_frame_ops *ops;             // latency form (orbx_frame_finish_begin / _end); its own handle: nobody else's calls interleave
    CamKey opsKey;
    const Frame *finishFor;          // the frame whose mvKeysUn / mGrid were filled by ExtractORB
    int finishN;
    orbx_frame_grid finishGrid;
    // Frame::ComputeBoW's vocabulary descent, begun on the device-resident descriptors the moment the left image is done (orbx_bow_job_begin);
    // Frame::ComputeBoW (shim/BoW_hip.cc) collects it through orbx_shim_early_bow_take - by frame id, the constructor's object is a temporary
    orbx_bow_job *bowJob;
    orbx_vocabulary *bowVoc;
    unsigned long long bowGen;
    long bowForId;                   // -1: none
    FrameAssist() : arrivals(0), pairFailed(0), stereo(0), stereoCap(0), stereoFor(0), ops(0), finishFor(0), finishN(0), bowJob(0), bowVoc(0), bowGen(0), bowForId(-1) { memset(&opsKey, 0, sizeof(opsKey)); memset(&finishGrid, 0, sizeof(finishGrid)); }
    ~FrameAssist() { if (stereo) orbx_matcher_destroy(stereo); if (ops) orbx_frame_ops_destroy(ops); if (bowJob) orbx_bow_job_destroy(bowJob); }
};
void FreeAssist(void *p) { delete (FrameAssist *)p; }
std::mutex gAssistMutex;
FrameAssist *AssistOf(ORBextractor *left)
{
    FrameAssist *a = (FrameAssist *)__atomic_load_n(&left->mpFrameAssist, __ATOMIC_ACQUIRE);
    if (a) return a;
    std::lock_guard<std::mutex> lock(gAssistMutex);
    a = (FrameAssist *)__atomic_load_n(&left->mpFrameAssist, __ATOMIC_ACQUIRE);
    if (!a) {
        a = new FrameAssist();
        left->mpFrameAssistFree = &FreeAssist;
        __atomic_store_n(&left->mpFrameAssist, (void *)a, __ATOMIC_RELEASE);
    }
    return a;
}
bool EarlyStart()
{
    const char *e = getenv("ORBX_SHIM_EARLY");      // (read per call: the tests switch it)
    return !(e && e[0] == '0');
}
bool EnsureStereoMatcher(FrameAssist *A, orbx_extractor *hl)
{
    const int need = orbx_extractor_capacity(hl);
    if (A->stereo && A->stereoCap >= need) return true;
    if (A->stereo) { orbx_matcher_destroy(A->stereo); A->stereo = 0; }
    if (orbx_matcher_create(orbx_shim::Device(), need, 1, &A->stereo) != ORBX_OK) { A->stereo = 0; return false; }
    A->stereoCap = need;
    return true;
}

struct PostCtx {
    Frame *F;
    int flag;
    FrameAssist *A;
    bool finishBegun;
    orbx_frame_grid grid;
};

// ORBextractor::SetPostExtract hook of Frame::ExtractORB: on the extractor's thread, the device has just finished this image (never throws)
void PostExtract(void *vc, bool ok)
{
    PostCtx *c = (PostCtx *)vc;
    Frame *F = c->F;
    FrameAssist *A = c->A;
    TRACE(c->flag == 0 ? "left: device done" : "right: device done");
    if (c->flag == 0 && ok && !Frame::mbInitialComputations) {
        orbx_camera cam;
        CamKey key;
        if (CameraOf(F->mK, F->mDistCoef, cam, key)) {
            if (A->ops && !(A->opsKey == key)) { orbx_frame_ops_destroy(A->ops); A->ops = 0; }
            if (!A->ops && orbx_frame_ops_create(orbx_shim::Device(), &cam, &A->ops) == ORBX_OK) A->opsKey = key;
            const orbx_frame_grid g = {Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
            if (A->ops && orbx_frame_finish_begin(A->ops, F->mpORBextractorLeft->Handle(), &g) == ORBX_OK) { c->finishBegun = true; c->grid = g; }
        }
    }
    if (c->flag == 0 && ok && F->mpORBvocabulary && orbx_shim_device_vocabulary) {      // Frame::ComputeBoW's descent: begun here, collected there
        unsigned long long gen = 0;
        orbx_vocabulary *dv = orbx_shim_device_vocabulary((const void *)F->mpORBvocabulary, &gen);
        if (dv) {
            if (A->bowJob && (A->bowVoc != dv || A->bowGen != gen)) { orbx_bow_job_destroy(A->bowJob); A->bowJob = 0; }
            if (!A->bowJob && orbx_bow_job_create(dv, &A->bowJob) == ORBX_OK) { A->bowVoc = dv; A->bowGen = gen; }
            if (A->bowJob && orbx_bow_job_begin(A->bowJob, F->mpORBextractorLeft->Handle(), 4) == ORBX_OK) A->bowForId = (long)F->mnId;
            TRACE("left: BoW descent launched");
        }
    }
    if (F->mpORBextractorRight) {      // the stereo constructor: two calls per frame, the second one to finish starts the match
        if (!ok) A->pairFailed.store(1);
        if (A->arrivals.fetch_add(1) == 1) {
            A->arrivals.store(0);
            const bool failed = A->pairFailed.exchange(0) != 0;
            orbx_extractor *hl = F->mpORBextractorLeft->Handle(), *hr = F->mpORBextractorRight->Handle();
            if (!failed && hl && hr && EnsureStereoMatcher(A, hl) && orbx_stereo_frame_begin(A->stereo, hl, hr, F->mbf, F->mb) == ORBX_OK) A->stereoFor = F;
            TRACE("stereo match launched");
        }
    }
}
}  // namespace

}  // namespace ORB_SLAM2

// for shim/BoW_hip.cc: the vocabulary job begun for frame `frameId` (nFeatures features) by the constructor that used `leftExtractor`, or NULL; one shot
extern "C" __attribute__((visibility("default"))) void *orbx_shim_early_bow_take(void *leftExtractor, long frameId, int nFeatures)
{
    if (!leftExtractor || !ORB_SLAM2::EarlyStart()) return 0;
    ORB_SLAM2::FrameAssist *A = (ORB_SLAM2::FrameAssist *)__atomic_load_n(&((ORB_SLAM2::ORBextractor *)leftExtractor)->mpFrameAssist, __ATOMIC_ACQUIRE);
    if (!A || !A->bowJob || A->bowForId < 0 || A->bowForId != frameId) return 0;
    A->bowForId = -1;
    (void)nFeatures;      // (checked by the caller against the job's own count)
    return A->bowJob;
}

namespace ORB_SLAM2
{

// ---------------------------------------------------------------------------------------------
//     void Frame::ExtractORB(int flag, const cv::Mat &im)                     src/Frame.cc:494-512
// The reference's body (one functor call) + the early start described above.
// ---------------------------------------------------------------------------------------------
static unsigned long gExtractCalls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_extract_orb_calls(void) { return gExtractCalls; }

void Frame::ExtractORB(int flag, const cv::Mat &im)
{
    __atomic_add_fetch(&gExtractCalls, 1, __ATOMIC_RELAXED);
    ShimTimer timer(P_EXTRACT);
    TRACE(flag == 0 ? "left: ExtractORB enters" : "right: ExtractORB enters");
    ORBextractor *ex = flag == 0 ? mpORBextractorLeft : mpORBextractorRight;
    PostCtx ctx = {this, flag, 0, false, {0.0f, 0.0f, 0.0f, 0.0f}};
    if (EarlyStart() && mpORBextractorLeft && ex) {
        ctx.A = AssistOf(mpORBextractorLeft);
        if (flag == 0) { ctx.A->finishFor = 0; ctx.A->stereoFor = 0; ctx.A->bowForId = -1; }      // (whatever an abandoned constructor left behind)
        ex->SetPostExtract(&PostExtract, &ctx);
    }
    if (mpORBextractorLeft && mpORBextractorRight)
        (flag == 0 ? mpORBextractorLeft : mpORBextractorRight)->ExpectPartner(flag == 0 ? mpORBextractorRight : mpORBextractorLeft);
    if (flag == 0) (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
    else (*mpORBextractorRight)(im, cv::Mat(), mvKeysRight, mDescriptorsRight);
    TRACE(flag == 0 ? "left: keypoints converted" : "right: keypoints converted");
    if (!ctx.finishBegun) return;
    // the left image's undistorted keypoints and grid, launched before the conversion above: mvKeysUn (src/Frame.cc:899-947) and mGrid (:460-491)
    FrameAssist *A = ctx.A;
    const orbx_keypoint *un = 0;
    const int32_t *off = 0, *idx = 0;
    int n = 0;
    mvKeysUn = mvKeys;      // (while the kernel runs; UndistortKeyPoints() rebuilds it if the early results are not used)
    if (orbx_frame_finish_end(A->ops, &un, &off, &idx, &n) != ORBX_OK || n != (int)mvKeys.size() || !off || (n > 0 && !idx)) return;
    TRACE("left: undistortion + grid arrived");
    if (un) for (int i = 0; i < n; i++) { mvKeysUn[(size_t)i].pt.x = un[i].x; mvKeysUn[(size_t)i].pt.y = un[i].y; }
    for (int x = 0; x < FRAME_GRID_COLS; x++)
        for (int y = 0; y < FRAME_GRID_ROWS; y++) {
            const int c = x * FRAME_GRID_ROWS + y;
            mGrid[x][y].assign(idx + off[c], idx + off[c + 1]);
        }
    A->finishN = n; A->finishGrid = ctx.grid; A->finishFor = this;
    TRACE("left: mvKeysUn + mGrid filled");
}

void Frame::ComputeStereoMatches()
{
    __atomic_add_fetch(&gStereoCalls, 1, __ATOMIC_RELAXED);
    ShimTimer timer(P_STEREO);
    TRACE("ComputeStereoMatches enters");
    mvuRight = std::vector<float>(N, -1.0f);   // :1029-1030
    mvDepth = std::vector<float>(N, -1.0f);
    TRACE("ComputeStereoMatches: vectors initialised");
    FrameAssist *A = AssistOf(mpORBextractorLeft);
    const bool begun = A->stereoFor == this;
    A->stereoFor = 0;
    if (N == 0) return;
    orbx_extractor *hl = mpORBextractorLeft->Handle(), *hr = mpORBextractorRight->Handle();
    if (begun) __atomic_add_fetch(&gEarlyStereo, 1, __ATOMIC_RELAXED);
    else {
        if (!EnsureStereoMatcher(A, hl) || orbx_stereo_frame_begin(A->stereo, hl, hr, mbf, mb) != ORBX_OK) {
            orbx_shim::Fail("Frame::ComputeStereoMatches");      // no stereo match: mvuRight / mvDepth stay -1 (:1029-1030)
            return;
        }
    }
    if (orbx_stereo_frame_end(A->stereo, &mvuRight[0], &mvDepth[0], N) != ORBX_OK) {
        orbx_shim::Fail("Frame::ComputeStereoMatches");
        mvuRight.assign((size_t)N, -1.0f); mvDepth.assign((size_t)N, -1.0f);
        return;
    }
    TRACE("ComputeStereoMatches returns");
}

// ---------------------------------------------------------------------------------------------
//     void Frame::UndistortKeyPoints()                    src/Frame.cc:899-947
//     void Frame::ComputeImageBounds(const cv::Mat &)     src/Frame.cc:950-1004
//     void Frame::AssignFeaturesToGrid()                  src/Frame.cc:460-491
// ---------------------------------------------------------------------------------------------
void Frame::UndistortKeyPoints()
{
    __atomic_add_fetch(&gUndistortCalls, 1, __ATOMIC_RELAXED);
    ShimTimer timer(P_UNDISTORT);
    if (EarlyStart() && mpORBextractorLeft) {
        const FrameAssist *A = AssistOf(mpORBextractorLeft);
        if (A->finishFor == this && A->finishN == N && (int)mvKeysUn.size() == N) return;      // filled by ExtractORB
    }
    if (mDistCoef.at<float>(0) == 0.0) {   // :901-905
        mvKeysUn = mvKeys;
        return;
    }
    std::vector<orbx_keypoint> in, out((size_t)(N > 0 ? N : 1));
    Pack(mvKeys, in);
    orbx_frame_ops *h = FrameOpsFor(mK, mDistCoef);
    {
        std::lock_guard<std::mutex> lock(gCallMutex);
        if (orbx_frame_undistort(h, N > 0 ? &in[0] : 0, N, &out[0]) != ORBX_OK) {
            orbx_shim::Fail("Frame::UndistortKeyPoints");
            mvKeysUn = mvKeys;      // (the vector must hold N entries for everything behind it; the distorted positions, as in :921-925)
            return;
        }
    }
    mvKeysUn.resize(N);   // :938-946
    for (int i = 0; i < N; i++) {
        cv::KeyPoint kp = mvKeys[i];
        kp.pt.x = out[(size_t)i].x;
        kp.pt.y = out[(size_t)i].y;
        mvKeysUn[i] = kp;
    }
}

void Frame::ComputeImageBounds(const cv::Mat &imLeft)
{
    __atomic_add_fetch(&gBoundsCalls, 1, __ATOMIC_RELAXED);
    ShimTimer timer(P_BOUNDS);
    float b[4];
    orbx_frame_ops *h = FrameOpsFor(mK, mDistCoef);
    std::lock_guard<std::mutex> lock(gCallMutex);
    if (orbx_frame_image_bounds(h, imLeft.cols, imLeft.rows, b) != ORBX_OK) {
        orbx_shim::Fail("Frame::ComputeImageBounds");
        b[0] = 0.0f; b[1] = (float)imLeft.cols; b[2] = 0.0f; b[3] = (float)imLeft.rows;      // the image itself (:986-991)
    }
    mnMinX = b[0]; mnMaxX = b[1]; mnMinY = b[2]; mnMaxY = b[3];
}

void Frame::AssignFeaturesToGrid()
{
    __atomic_add_fetch(&gGridCalls, 1, __ATOMIC_RELAXED);
    ShimTimer timer(P_GRID);
    const orbx_frame_grid g = {mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv};   // what PosInGrid reads, :868-878
    if (EarlyStart() && mpORBextractorLeft) {
        FrameAssist *A = AssistOf(mpORBextractorLeft);
        const bool filled = A->finishFor == this && A->finishN == N && memcmp(&A->finishGrid, &g, sizeof(g)) == 0;
        A->finishFor = 0;
        if (filled) { __atomic_add_fetch(&gEarlyFills, 1, __ATOMIC_RELAXED); TRACE("AssignFeaturesToGrid: already filled"); return; }      // by ExtractORB, from the same statics
    }
    std::vector<orbx_keypoint> in;
    Pack(mvKeysUn, in);
    std::vector<int32_t> off((size_t)FRAME_GRID_COLS * FRAME_GRID_ROWS + 1), idx((size_t)(N > 0 ? N : 1));
    orbx_frame_ops *h = FrameOpsFor(mK, mDistCoef);
    {
        std::lock_guard<std::mutex> lock(gCallMutex);
        if (orbx_frame_assign_grid(h, &g, N > 0 ? &in[0] : 0, N, &off[0], &idx[0]) != ORBX_OK) {
            orbx_shim::Fail("Frame::AssignFeaturesToGrid");      // the grid stays empty
            return;
        }
    }
    for (int x = 0; x < FRAME_GRID_COLS; x++)
        for (int y = 0; y < FRAME_GRID_ROWS; y++) {
            const int c = x * FRAME_GRID_ROWS + y;
            mGrid[x][y].assign(idx.begin() + off[(size_t)c], idx.begin() + off[(size_t)c + 1]);
        }
}

}  // namespace ORB_SLAM2

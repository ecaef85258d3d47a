// shim/Frame_hip.cc -- HIP bodies for ORB_SLAM2::Frame::ComputeStereoMatches, UndistortKeyPoints,
// ComputeImageBounds and AssignFeaturesToGrid (+ Frame::ExtractORB with the stereo pair hint).
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

#include <map>
#include <mutex>

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

static unsigned long gStereoCalls = 0, gUndistortCalls = 0, gBoundsCalls = 0, gGridCalls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_undistort_calls(void) { return gUndistortCalls; }
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_image_bounds_calls(void) { return gBoundsCalls; }
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_assign_grid_calls(void) { return gGridCalls; }
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_compute_stereo_matches_calls(void) { return gStereoCalls; }

namespace ORB_SLAM2
{

namespace
{
struct ThreadStereo {
    orbx_matcher *h;
    int cap;
    ThreadStereo() : h(0), cap(0) {}
    ~ThreadStereo() { if (h) orbx_matcher_destroy(h); }
};
thread_local ThreadStereo tStereo;
}  // namespace

// ---------------------------------------------------------------------------------------------
//     void Frame::ExtractORB(int flag, const cv::Mat &im)                     src/Frame.cc:494-512
// The reference's body, plus one line: in the stereo constructor the two calls run on two threads (src/Frame.cc:159-167), and each
// tells liborbx that the other extractor's call is on its way, so that both frames run as ONE launch set (include/orbx.h:
// orbx_extractor_expect_partner).  Monocular / RGB-D frames (mpORBextractorRight == NULL, :283, :394) give no hint and never wait.
// ---------------------------------------------------------------------------------------------
static unsigned long gExtractCalls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_extract_orb_calls(void) { return gExtractCalls; }

void Frame::ExtractORB(int flag, const cv::Mat &im)
{
    __atomic_add_fetch(&gExtractCalls, 1, __ATOMIC_RELAXED);
    ShimTimer timer(P_EXTRACT);
    if (mpORBextractorLeft && mpORBextractorRight)
        (flag == 0 ? mpORBextractorLeft : mpORBextractorRight)->ExpectPartner(flag == 0 ? mpORBextractorRight : mpORBextractorLeft);
    if (flag == 0) (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
    else (*mpORBextractorRight)(im, cv::Mat(), mvKeysRight, mDescriptorsRight);
}

void Frame::ComputeStereoMatches()
{
    __atomic_add_fetch(&gStereoCalls, 1, __ATOMIC_RELAXED);
    ShimTimer timer(P_STEREO);
    mvuRight = std::vector<float>(N, -1.0f);   // :1029-1030
    mvDepth = std::vector<float>(N, -1.0f);
    if (N == 0) return;
    orbx_extractor *hl = mpORBextractorLeft->Handle(), *hr = mpORBextractorRight->Handle();
    const int need = orbx_extractor_capacity(hl);
    if (!tStereo.h || tStereo.cap < need) {
        if (tStereo.h) { orbx_matcher_destroy(tStereo.h); tStereo.h = 0; }
        if (orbx_matcher_create(0, need, 1, &tStereo.h) != ORBX_OK)
            throw std::runtime_error(std::string("Frame::ComputeStereoMatches (orbx): ") + orbx_last_error());
        tStereo.cap = need;
    }
    if (orbx_stereo_frame(tStereo.h, hl, hr, mbf, mb, &mvuRight[0], &mvDepth[0], N) != ORBX_OK)
        throw std::runtime_error(std::string("Frame::ComputeStereoMatches (orbx): ") + orbx_last_error());
}

// ---------------------------------------------------------------------------------------------
//     void Frame::UndistortKeyPoints()                    src/Frame.cc:899-947
//     void Frame::ComputeImageBounds(const cv::Mat &)     src/Frame.cc:950-1004
//     void Frame::AssignFeaturesToGrid()                  src/Frame.cc:460-491
// One device handle per camera (mK, mDistCoef), shared by the threads that build frames.
// ---------------------------------------------------------------------------------------------
namespace
{
struct CamKey {
    float v[10];
    bool operator<(const CamKey &o) const
    {
        for (int i = 0; i < 10; i++) if (v[i] != o.v[i]) return v[i] < o.v[i];
        return false;
    }
};
std::mutex gOpsMutex;
std::map<CamKey, orbx_frame_ops *> gOps;

orbx_frame_ops *FrameOpsFor(const cv::Mat &K, const cv::Mat &D)
{
    orbx_camera cam;
    cam.fx = K.at<float>(0, 0); cam.fy = K.at<float>(1, 1); cam.cx = K.at<float>(0, 2); cam.cy = K.at<float>(1, 2);
    cam.ndist = D.rows * D.cols;
    if (cam.ndist != 4 && cam.ndist != 5) throw std::runtime_error("Frame (orbx): mDistCoef must hold 4 or 5 coefficients");
    for (int i = 0; i < 5; i++) cam.dist[i] = i < cam.ndist ? D.at<float>(i) : 0.0f;
    CamKey key = {{cam.fx, cam.fy, cam.cx, cam.cy, cam.dist[0], cam.dist[1], cam.dist[2], cam.dist[3], cam.dist[4], (float)cam.ndist}};
    std::lock_guard<std::mutex> lock(gOpsMutex);
    std::map<CamKey, orbx_frame_ops *>::iterator it = gOps.find(key);
    if (it != gOps.end()) return it->second;
    orbx_frame_ops *h = 0;
    if (orbx_frame_ops_create(0, &cam, &h) != ORBX_OK) throw std::runtime_error(std::string("Frame (orbx): ") + orbx_last_error());
    gOps[key] = h;
    return h;
}

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
std::mutex gCallMutex;   // a handle is not re-entrant (include/orbx.h); frames are built by one thread at a time in the reference
}  // namespace

void Frame::UndistortKeyPoints()
{
    __atomic_add_fetch(&gUndistortCalls, 1, __ATOMIC_RELAXED);
    ShimTimer timer(P_UNDISTORT);
    if (mDistCoef.at<float>(0) == 0.0) {   // :901-905
        mvKeysUn = mvKeys;
        return;
    }
    std::vector<orbx_keypoint> in, out((size_t)(N > 0 ? N : 1));
    Pack(mvKeys, in);
    orbx_frame_ops *h = FrameOpsFor(mK, mDistCoef);
    {
        std::lock_guard<std::mutex> lock(gCallMutex);
        if (orbx_frame_undistort(h, N > 0 ? &in[0] : 0, N, &out[0]) != ORBX_OK)
            throw std::runtime_error(std::string("Frame::UndistortKeyPoints (orbx): ") + orbx_last_error());
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
    if (orbx_frame_image_bounds(h, imLeft.cols, imLeft.rows, b) != ORBX_OK)
        throw std::runtime_error(std::string("Frame::ComputeImageBounds (orbx): ") + orbx_last_error());
    mnMinX = b[0]; mnMaxX = b[1]; mnMinY = b[2]; mnMaxY = b[3];
}

void Frame::AssignFeaturesToGrid()
{
    __atomic_add_fetch(&gGridCalls, 1, __ATOMIC_RELAXED);
    ShimTimer timer(P_GRID);
    std::vector<orbx_keypoint> in;
    Pack(mvKeysUn, in);
    std::vector<int32_t> off((size_t)FRAME_GRID_COLS * FRAME_GRID_ROWS + 1), idx((size_t)(N > 0 ? N : 1));
    const orbx_frame_grid g = {mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv};   // what PosInGrid reads, :868-878
    orbx_frame_ops *h = FrameOpsFor(mK, mDistCoef);
    {
        std::lock_guard<std::mutex> lock(gCallMutex);
        if (orbx_frame_assign_grid(h, &g, N > 0 ? &in[0] : 0, N, &off[0], &idx[0]) != ORBX_OK)
            throw std::runtime_error(std::string("Frame::AssignFeaturesToGrid (orbx): ") + orbx_last_error());
    }
    for (int x = 0; x < FRAME_GRID_COLS; x++)
        for (int y = 0; y < FRAME_GRID_ROWS; y++) {
            const int c = x * FRAME_GRID_ROWS + y;
            mGrid[x][y].assign(idx.begin() + off[(size_t)c], idx.begin() + off[(size_t)c + 1]);
        }
}

}  // namespace ORB_SLAM2

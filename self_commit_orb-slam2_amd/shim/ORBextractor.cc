// shim/ORBextractor.cc -- ORB_SLAM2::ORBextractor over liborbx (see ORBextractor.h).
//
// Error conventions follow the reference (src/ORBextractor.cc:1544-1668): empty image ->
// silent return, outputs untouched; no keypoints -> descriptors.release().  The reference's
// constructor and functor never throw and have no error channel: a device error here is
// written to std::cerr (as the reference reports its own failures, e.g. src/System.cc:61), counted
// (ErrorCount / LastError) and the call returns with EMPTY outputs - not with whatever the
// caller's vectors held from the previous frame.  sbThrowOnError / ORBX_SHIM_FATAL=1: throw instead.
#include "ORBextractor.h"

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>

#include "orbx.h"

// Defined (non-zero) by shim/Frame_hip.cc: Frame::ComputeStereoMatches then reads the pyramid on the
// device.  When that file is NOT linked - a build that swaps only the extractor and keeps the reference's
// own ComputeStereoMatches, which reads mvImagePyramid (src/Frame.cc:1044,1248,1272,1281) - the symbol is
// absent and every operator() refills the public member, as the reference's does.
extern "C" __attribute__((weak)) int orbx_shim_device_stereo_linked;

namespace ORB_SLAM2
{

static int gDevice = 0;
void ORBextractor::SetDevice(int device) { gDevice = device; }

static bool EnvFatal() { const char *e = getenv("ORBX_SHIM_FATAL"); return e && e[0] == '1'; }
bool ORBextractor::sbThrowOnError = EnvFatal();

bool ORBextractor::Fail(const char *what)
{
    mLastError = std::string(what) + " failed: " + orbx_last_error();
    ++mnErrors;
    std::cerr << "ORBextractor (orbx): " << mLastError << std::endl;
    if (sbThrowOnError) throw std::runtime_error("ORBextractor (orbx): " + mLastError);
    return false;
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : mbKeepHostPyramid(!(&orbx_shim_device_stereo_linked && orbx_shim_device_stereo_linked)), mpFrameAssist(0), mpFrameAssistFree(0), nfeatures(_nfeatures), scaleFactor(_scaleFactor),
      nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST), mpHandle(0), mMaxW(0), mMaxH(0), mLastW(0), mLastH(0), mPostFn(0), mPostCtx(0), mnErrors(0),
      mbDead(false)
{
    mvImagePyramid.resize(nlevels);
    // The tables come from the library so that getters and kernels can never disagree (no device needed for them).
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    orbx_extractor_config cfg = orbx_extractor_config();
    cfg.nfeatures = nfeatures; cfg.scale_factor = (float)scaleFactor; cfg.nlevels = nlevels;
    if (nlevels > 0 && orbx_extractor_tables_for(&cfg, &mvScaleFactor[0], &mvInvScaleFactor[0], &mvLevelSigma2[0], &mvInvLevelSigma2[0], &mnFeaturesPerLevel[0]) != ORBX_OK)
        Fail("tables");
    EnsureHandle(640, 480);
}

ORBextractor::~ORBextractor()
{
    if (mpFrameAssist && mpFrameAssistFree) mpFrameAssistFree(mpFrameAssist);      // (its handles read this extractor's buffers: first)
    if (mpHandle) orbx_extractor_destroy(mpHandle);
}

bool ORBextractor::EnsureHandle(int width, int height)
{
    if (mpHandle && width <= mMaxW && height <= mMaxH) return true;
    if (mbDead) return false;          // no device at construction: counted once per call in operator(), not re-opened per frame
    if (mpHandle) { orbx_extractor_destroy(mpHandle); mpHandle = 0; }
    orbx_extractor_config cfg = orbx_extractor_config();
    cfg.nfeatures = nfeatures; cfg.scale_factor = (float)scaleFactor; cfg.nlevels = nlevels;
    cfg.ini_th_fast = iniThFAST; cfg.min_th_fast = minThFAST;
    cfg.max_width = width > mMaxW ? width : mMaxW;
    cfg.max_height = height > mMaxH ? height : mMaxH;
    cfg.max_batch = 1; cfg.device = gDevice;
    const int rc = orbx_extractor_create(&cfg, &mpHandle);
    if (rc != ORBX_OK) { mpHandle = 0; mbDead = rc == ORBX_ERR_NODEVICE; return Fail("create"); }
    mMaxW = cfg.max_width; mMaxH = cfg.max_height;
    return true;
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors)
{
    // the one-shot hook of this call (SetPostExtract): taken now, run exactly once on every path out of the device call
    struct PostHook {
        PostExtractFn fn; void *ctx; bool ran;
        void run(bool ok) { if (fn && !ran) { ran = true; fn(ctx, ok); } }
        ~PostHook() { run(false); }
    } post = {mPostFn, mPostCtx, false};
    mPostFn = 0; mPostCtx = 0;
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    if (!EnsureHandle(image.cols, image.rows)) {
        _keypoints.clear(); _descriptors.release();
        if (mbDead) { ++mnErrors; if (sbThrowOnError) throw std::runtime_error("ORBextractor (orbx): " + mLastError); }
        return;
    }

    // results arrive in the handle's pinned buffer; they are converted straight from there
    const orbx_keypoint *kps = 0;
    const unsigned char *desc = 0;
    int n = 0;
    orbx_host_pyramid pyr;
    if (orbx_extract_view_pyramid(mpHandle, image.data, image.cols, image.rows, (int)image.step, &kps, &desc, &n, mbKeepHostPyramid ? &pyr : 0) != ORBX_OK) {
        _keypoints.clear(); _descriptors.release();       // never the previous frame's data
        Fail("extract");
        return;
    }
    post.run(true);

    if (n == 0)
        _descriptors.release();
    else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        if (d.isContinuous()) memcpy(d.data, desc, (size_t)n * 32);
        else for (int i = 0; i < n; i++) memcpy(d.ptr(i), desc + (size_t)i * 32, 32);
    }
    // orbx_keypoint IS cv::KeyPoint's layout (pt.x, pt.y, size, angle, response, octave, class_id: 28 bytes, checked below): the vector is built
    // straight from the pinned buffer in one pass (resize + a field-by-field loop cost 10 of the call's 14 host microseconds at 2000 keypoints)
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint) && offsetof(cv::KeyPoint, pt) == offsetof(orbx_keypoint, x) &&
                      offsetof(cv::KeyPoint, size) == offsetof(orbx_keypoint, size) && offsetof(cv::KeyPoint, angle) == offsetof(orbx_keypoint, angle) &&
                      offsetof(cv::KeyPoint, response) == offsetof(orbx_keypoint, response) && offsetof(cv::KeyPoint, octave) == offsetof(orbx_keypoint, octave) &&
                      offsetof(cv::KeyPoint, class_id) == offsetof(orbx_keypoint, class_id),
                  "cv::KeyPoint and orbx_keypoint must have the same layout");
    {
        const cv::KeyPoint *first = reinterpret_cast<const cv::KeyPoint *>(kps);
        _keypoints.assign(first, first + n);
    }
    mLastW = image.cols; mLastH = image.rows;
    if (mbKeepHostPyramid) {
        // the pyramid came back with the results (same launch set, same wait): the public member becomes VIEWS of the handle's pinned
        // memory - level 0 is the staged copy of `image` -, valid until the next call, which is how long the reference's member holds a
        // frame's pyramid too (ComputePyramid overwrites it, src/ORBextractor.cc:1680-1733)
        for (int level = 0; level < nlevels && level < pyr.nlevels; ++level)
            mvImagePyramid[level] = cv::Mat(pyr.height[level], pyr.width[level], CV_8UC1, (void *)pyr.level[level], (size_t)pyr.stride[level]);
    }
}

void ORBextractor::ExpectPartner(ORBextractor *other)
{
    if (mpHandle) orbx_extractor_expect_partner(mpHandle, other ? other->mpHandle : 0);
}

void ORBextractor::DownloadImagePyramid()
{
    if (!mpHandle || mLastW <= 0) return;
    std::vector<unsigned char *> ptr((size_t)nlevels);
    std::vector<int> step((size_t)nlevels);
    for (int level = 0; level < nlevels; ++level) {
        int w = 0, h = 0;
        orbx_pyramid_level_size(mpHandle, mLastW, mLastH, level, &w, &h);
        mvImagePyramid[level].release();             // (a view of the handle's pinned memory from the last call: an owning copy now)
        mvImagePyramid[level].create(h, w, CV_8UC1);
        ptr[(size_t)level] = mvImagePyramid[level].data;
        step[(size_t)level] = (int)mvImagePyramid[level].step;
    }
    if (orbx_download_pyramid_all(mpHandle, 0, &ptr[0], &step[0], nlevels) != ORBX_OK) Fail("pyramid");
}

} // namespace ORB_SLAM2

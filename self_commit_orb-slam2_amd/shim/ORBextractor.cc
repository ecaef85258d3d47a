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
#include "shim_error.h"

// Defined (non-zero) by shim/Frame_hip.cc: Frame::ComputeStereoMatches then reads the pyramid on the
// device.  When that file is NOT linked - a build that swaps only the extractor and keeps the reference's
// own ComputeStereoMatches, which reads mvImagePyramid (src/Frame.cc:1044,1248,1272,1281) - the symbol is
// absent and every operator() refills the public member, as the reference's does.
extern "C" __attribute__((weak)) int orbx_shim_device_stereo_linked;

namespace ORB_SLAM2
{

static int gDevice = 0;
void ORBextractor::SetDevice(int device) { gDevice = device; orbx_shim::Device() = device; }

static bool EnvFatal() { const char *e = getenv("ORBX_SHIM_FATAL"); return e && e[0] == '1'; }
bool ORBextractor::sbThrowOnError = EnvFatal();

bool ORBextractor::Fail(const char *what)
{
    mLastError = std::string(what) + " failed: " + orbx_last_error();
    ++mnErrors;
    {   // the process-wide channel of all shim files (shim_error.h)
        orbx_shim::ErrorState &e = orbx_shim::Errors();
        std::lock_guard<std::mutex> lock(e.m);
        e.last = "ORBextractor (orbx): " + mLastError;
        e.count.fetch_add(1);
    }
    std::cerr << "ORBextractor (orbx): " << mLastError << std::endl;
    if (sbThrowOnError) throw std::runtime_error("ORBextractor (orbx): " + mLastError);
    return false;
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : mbKeepHostPyramid(!(&orbx_shim_device_stereo_linked && orbx_shim_device_stereo_linked)), mbViewHostPyramid(false), mpFrameAssist(0), mpFrameAssistFree(0), nfeatures(_nfeatures), scaleFactor(_scaleFactor),
      nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST), mpHandle(0), mMaxW(0), mMaxH(0), mLastW(0), mLastH(0), mPostFn(0), mPostCtx(0), mnErrors(0),
      mbDead(false), mPyrState(PYR_NONE), mbPyrRead(false), mPyrLevels(0)
{
    mvImagePyramid.resize(nlevels);
    mvImagePyramid.mpOwner = this;
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
    if (mPyrState == PYR_VIEWS) DropImagePyramid();
    mPyrState = PYR_NONE;          // (the levels the caller may still hold are owning copies: nothing of theirs lives in the handle)
    if (mpHandle) orbx_extractor_destroy(mpHandle);
}

bool ORBextractor::EnsureHandle(int width, int height)
{
    if (mpHandle && width <= mMaxW && height <= mMaxH) return true;
    if (mbDead) return false;          // no device at construction: counted once per call in operator(), not re-opened per frame
    if (mpHandle) {
        if (mPyrState == PYR_VIEWS) DropImagePyramid();      // (opt-in views of the handle's pinned memory: emptied, never left dangling)
        if (mPyrState == PYR_IN_PINNED || mPyrState == PYR_ON_DEVICE) { const bool v = mbViewHostPyramid; mbViewHostPyramid = false; FillImagePyramid(); mbViewHostPyramid = v; }      // the last frame's pyramid leaves the handle before the handle goes
        orbx_extractor_destroy(mpHandle); mpHandle = 0;
    }
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
        _keypoints.clear(); _descriptors.release(); DropImagePyramid();
        if (mbDead) { ++mnErrors; if (sbThrowOnError) throw std::runtime_error("ORBextractor (orbx): " + mLastError); }
        return;
    }

    // results arrive in the handle's pinned buffer; they are converted straight from there
    const orbx_keypoint *kps = 0;
    const unsigned char *desc = 0;
    int n = 0;
    orbx_host_pyramid pyr;
    // the pyramid rides along when somebody read the previous frame's (the reference's stereo ComputeStereoMatches does, for every frame)
    const bool bringPyramid = mbKeepHostPyramid && mbPyrRead;
    mbPyrRead = false;
    if (mPyrState == PYR_VIEWS) DropImagePyramid();       // (the pinned memory behind the opt-in views is about to be rewritten)
    if (orbx_extract_view_pyramid(mpHandle, image.data, image.cols, image.rows, (int)image.step, &kps, &desc, &n, bringPyramid ? &pyr : 0) != ORBX_OK) {
        _keypoints.clear(); _descriptors.release();       // never the previous frame's data
        DropImagePyramid();
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
    // where this frame's pyramid is, for the first reader of mvImagePyramid (FillImagePyramid): in the handle's pinned memory (it came back with
    // the results: same launch set, same wait), or still on the device only
    if (bringPyramid) {
        mPyrLevels = pyr.nlevels < nlevels ? pyr.nlevels : nlevels;
        for (int level = 0; level < mPyrLevels; ++level) {
            mPyrLevel[level] = pyr.level[level]; mPyrW[level] = pyr.width[level]; mPyrH[level] = pyr.height[level]; mPyrStride[level] = pyr.stride[level];
        }
        mPyrState = PYR_IN_PINNED;
    } else mPyrState = PYR_ON_DEVICE;
}

void ORBextractor::DropImagePyramid()
{
    for (size_t level = 0; level < mvImagePyramid.mv.size(); ++level) mvImagePyramid.mv[level] = cv::Mat();
    mPyrState = PYR_NONE;
}

// First access of mvImagePyramid after a call: every level a fresh owning cv::Mat (the reference assigns new Mats per call,
// src/ORBextractor.cc:1687-1689: a level somebody kept from an earlier frame is not touched).
void ORBextractor::FillImagePyramid()
{
    if (mPyrState == PYR_NONE || mPyrState == PYR_OWNED || mPyrState == PYR_VIEWS) return;
    mbPyrRead = true;
    if (mPyrState == PYR_IN_PINNED && mbViewHostPyramid) {
        for (int level = 0; level < mPyrLevels; ++level)
            mvImagePyramid.mv[(size_t)level] = cv::Mat(mPyrH[level], mPyrW[level], CV_8UC1, (void *)mPyrLevel[level], (size_t)mPyrStride[level]);
        mPyrState = PYR_VIEWS;
        return;
    }
    if (mPyrState == PYR_IN_PINNED) {
        for (int level = 0; level < mPyrLevels; ++level) {
            cv::Mat m(mPyrH[level], mPyrW[level], CV_8UC1);
            for (int y = 0; y < mPyrH[level]; y++) memcpy(m.ptr(y), mPyrLevel[level] + (size_t)y * (size_t)mPyrStride[level], (size_t)mPyrW[level]);
            mvImagePyramid.mv[(size_t)level] = m;
        }
        mPyrState = PYR_OWNED;
        return;
    }
    mPyrState = PYR_OWNED;              // (whatever the download below does, it is not retried per access)
    DownloadImagePyramid();
}

void ORBextractor::ExpectPartner(ORBextractor *other)
{
    if (mpHandle) orbx_extractor_expect_partner(mpHandle, other ? other->mpHandle : 0);
}

void ORBextractor::DownloadImagePyramid()
{
    if (!mpHandle || mLastW <= 0) return;
    if (mPyrState == PYR_IN_PINNED) { const bool v = mbViewHostPyramid; mbViewHostPyramid = false; FillImagePyramid(); mbViewHostPyramid = v; return; }      // (owning copies, whatever the flag)
    std::vector<unsigned char *> ptr((size_t)nlevels);
    std::vector<int> step((size_t)nlevels);
    for (int level = 0; level < nlevels; ++level) {
        int w = 0, h = 0;
        orbx_pyramid_level_size(mpHandle, mLastW, mLastH, level, &w, &h);
        mvImagePyramid.mv[(size_t)level] = cv::Mat(h, w, CV_8UC1);      // (a fresh owning level, see FillImagePyramid)
        ptr[(size_t)level] = mvImagePyramid.mv[(size_t)level].data;
        step[(size_t)level] = (int)mvImagePyramid.mv[(size_t)level].step;
    }
    mPyrState = PYR_OWNED;
    mbPyrRead = true;
    if (orbx_download_pyramid_all(mpHandle, 0, &ptr[0], &step[0], nlevels) != ORBX_OK) { DropImagePyramid(); Fail("pyramid"); }
}

} // namespace ORB_SLAM2

// shim/ORBextractor.cc -- ORB_SLAM2::ORBextractor over liborbx (see ORBextractor.h).
//
// Error conventions follow the reference (src/ORBextractor.cc:1544-1668): empty image ->
// silent return, outputs untouched; no keypoints -> descriptors.release().  A device error
// cannot be reported through the void functor, so it throws std::runtime_error with
// orbx_last_error() (the reference would have asserted / crashed in the same situation).
#include "ORBextractor.h"

#include <stdexcept>
#include <string>

#include "orbx.h"

namespace ORB_SLAM2
{

static int gDevice = 0;
void ORBextractor::SetDevice(int device) { gDevice = device; }

static void Fail(const char *what)
{
    throw std::runtime_error(std::string("ORBextractor (orbx): ") + what + ": " + orbx_last_error());
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : mbKeepHostPyramid(false), nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST),
      minThFAST(_minThFAST), mpHandle(0), mMaxW(0), mMaxH(0), mLastW(0), mLastH(0)
{
    mvImagePyramid.resize(nlevels);
    // The tables come from the library so that getters and kernels can never disagree.
    EnsureHandle(640, 480);
}

ORBextractor::~ORBextractor()
{
    if (mpHandle) orbx_extractor_destroy(mpHandle);
}

void ORBextractor::EnsureHandle(int width, int height)
{
    if (mpHandle && width <= mMaxW && height <= mMaxH) return;
    if (mpHandle) { orbx_extractor_destroy(mpHandle); mpHandle = 0; }
    orbx_extractor_config cfg = orbx_extractor_config();
    cfg.nfeatures = nfeatures; cfg.scale_factor = (float)scaleFactor; cfg.nlevels = nlevels;
    cfg.ini_th_fast = iniThFAST; cfg.min_th_fast = minThFAST;
    cfg.max_width = width > mMaxW ? width : mMaxW;
    cfg.max_height = height > mMaxH ? height : mMaxH;
    cfg.max_batch = 1; cfg.device = gDevice;
    if (orbx_extractor_create(&cfg, &mpHandle) != ORBX_OK) Fail("create");
    mMaxW = cfg.max_width; mMaxH = cfg.max_height;
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    int nl = 0;
    if (orbx_extractor_tables(mpHandle, &nl, &mvScaleFactor[0], &mvInvScaleFactor[0], &mvLevelSigma2[0], &mvInvLevelSigma2[0],
                              &mnFeaturesPerLevel[0]) != ORBX_OK)
        Fail("tables");
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors)
{
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    EnsureHandle(image.cols, image.rows);

    const int cap = orbx_extractor_capacity(mpHandle);
    std::vector<orbx_keypoint> kps((size_t)cap);
    std::vector<unsigned char> desc((size_t)cap * 32);
    int n = 0;
    if (orbx_extract(mpHandle, image.data, image.cols, image.rows, (int)image.step, &kps[0], &desc[0], cap, &n) != ORBX_OK) Fail("extract");

    if (n == 0)
        _descriptors.release();
    else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        if (d.isContinuous()) memcpy(d.data, &desc[0], (size_t)n * 32);
        else for (int i = 0; i < n; i++) memcpy(d.ptr(i), &desc[(size_t)i * 32], 32);
    }
    _keypoints.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        const orbx_keypoint &k = kps[(size_t)i];
        cv::KeyPoint &o = _keypoints[(size_t)i];
        o.pt.x = k.x; o.pt.y = k.y; o.size = k.size; o.angle = k.angle; o.response = k.response; o.octave = k.octave; o.class_id = k.class_id;
    }
    mLastW = image.cols; mLastH = image.rows;
    if (mbKeepHostPyramid) DownloadImagePyramid();
}

// mvImagePyramid of the last frame, on demand: the pyramid stays on the device (where shim/Frame_hip.cc's ComputeStereoMatches reads
// it) until the next operator() call.
void ORBextractor::DownloadImagePyramid()
{
    if (!mpHandle || mLastW <= 0) return;
    std::vector<unsigned char *> ptr((size_t)nlevels);
    std::vector<int> step((size_t)nlevels);
    for (int level = 0; level < nlevels; ++level) {
        int w = 0, h = 0;
        orbx_pyramid_level_size(mpHandle, mLastW, mLastH, level, &w, &h);
        mvImagePyramid[level].create(h, w, CV_8UC1);
        ptr[(size_t)level] = mvImagePyramid[level].data;
        step[(size_t)level] = (int)mvImagePyramid[level].step;
    }
    if (orbx_download_pyramid_all(mpHandle, 0, &ptr[0], &step[0], nlevels) != ORBX_OK) Fail("pyramid");
}

} // namespace ORB_SLAM2

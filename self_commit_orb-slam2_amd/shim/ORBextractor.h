// shim/ORBextractor.h -- drop-in replacement header for the reference's include/ORBextractor.h.
//
// Same namespace, class name, constructor, functor signature, getters and the public
// `mvImagePyramid` member as ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:92-161),
// so src/Frame.cc, src/Tracking.cc and src/Frame.cc::ComputeStereoMatches compile and behave
// unchanged; the body marshals to the C ABI of liborbx.so (include/orbx.h) and the work runs
// on the MI355X.  No CPU fallback: construction throws std::runtime_error without a GPU.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <vector>

#include <opencv/cv.h>

struct orbx_extractor;

namespace ORB_SLAM2
{

class ORBextractor
{
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();

    // Computes the ORB features and descriptors of an image; `mask` is ignored, exactly like
    // the reference implementation does.
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Host copy of the image pyramid of the last frame.  The reference's only reader is Frame::ComputeStereoMatches
    // (src/Frame.cc:1044,1248).  Safe by default: mbKeepHostPyramid starts TRUE - every call refills the member with one ~1 MB
    // device->host transfer (~0.1 ms), exactly what a build that swaps only the extractor needs - and starts FALSE only when
    // shim/Frame_hip.cc is linked into the same binary (it defines orbx_shim_device_stereo_linked): that ComputeStereoMatches reads
    // the DEVICE pyramid.  Anything else that wants the images then calls DownloadImagePyramid() when it does.
    std::vector<cv::Mat> mvImagePyramid;
    bool mbKeepHostPyramid;
    void DownloadImagePyramid();

    // Device used by extractors constructed afterwards (default 0).
    static void SetDevice(int device);
    // liborbx handle holding the device-resident results and pyramid of the last frame
    // (consumed by the HIP body of Frame::ComputeStereoMatches, shim/Frame_hip.cc).
    orbx_extractor *Handle() const { return mpHandle; }

private:
    ORBextractor(const ORBextractor &);
    ORBextractor &operator=(const ORBextractor &);
    bool EnsureHandle(int width, int height);

    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel;
    orbx_extractor *mpHandle;
    int mMaxW, mMaxH, mLastW, mLastH;
};

} // namespace ORB_SLAM2

#endif

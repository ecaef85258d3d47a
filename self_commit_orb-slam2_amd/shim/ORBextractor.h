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
    // (src/Frame.cc:1044,1248); with shim/Frame_hip.cc linked that function reads the DEVICE pyramid, so the copy is off by
    // default.  A build that keeps the reference's own ComputeStereoMatches sets mbKeepHostPyramid = true (every call then ends
    // with one ~1 MB device->host transfer), anything else that wants the images calls DownloadImagePyramid() when it does.
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
    void EnsureHandle(int width, int height);

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

// shim/ORBextractor.h -- drop-in replacement header for the reference's include/ORBextractor.h.
//
// Same namespace, class name, constructor, functor signature, getters and the public
// `mvImagePyramid` member as ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:92-161),
// so src/Frame.cc, src/Tracking.cc and src/Frame.cc::ComputeStereoMatches compile and behave
// unchanged; the body marshals to the C ABI of liborbx.so (include/orbx.h) and the work runs
// on the MI355X.  No CPU fallback, and - like the reference's constructor and functor - nothing throws by default: without a GPU, or
// on a device error, operator() returns EMPTY outputs, counts the failure (ErrorCount / LastError) and writes it to std::cerr;
// sbThrowOnError = true (or ORBX_SHIM_FATAL=1 in the environment) turns every such failure into a std::runtime_error instead.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <string>
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
    // (src/Frame.cc:1044,1248).  Safe by default: mbKeepHostPyramid starts TRUE - every call leaves the frame's pyramid in the member,
    // as VIEWS of the handle's pinned memory that the launch set itself filled (levels >= 1; level 0 is the staged copy of the input):
    // no second transfer, no second wait, no host copy; valid until the next call on this extractor -, exactly what a build that swaps
    // only the extractor needs - and starts FALSE only when shim/Frame_hip.cc is linked into the same binary (it defines
    // orbx_shim_device_stereo_linked): that ComputeStereoMatches reads the DEVICE pyramid.  Anything else that wants the images then
    // calls DownloadImagePyramid() when it does (owning copies).
    std::vector<cv::Mat> mvImagePyramid;
    bool mbKeepHostPyramid;
    void DownloadImagePyramid();

    // One-shot hint for the next operator() call: `other`'s call is about to arrive on another thread (the stereo Frame constructor runs the
    // left and the right extractor on two threads, src/Frame.cc:159-167); with ORBX_COMBINE_PARTNER_US set, liborbx runs both frames as one
    // launch set (off by default: include/orbx.h says why).  Set by the HIP body of Frame::ExtractORB (shim/Frame_hip.cc).
    void ExpectPartner(ORBextractor *other);

    // One-shot hook for the next operator() call: run on the calling thread when the device has finished the frame (ok) or the call has
    // failed (!ok), BEFORE the results are converted into the caller's containers - whatever the hook launches overlaps that conversion.
    // shim/Frame_hip.cc starts the rest of the Frame constructor from it (UndistortKeyPoints / AssignFeaturesToGrid of the left image,
    // ComputeStereoMatches once both images are done).  Always called exactly once per operator() call it was set for.
    typedef void (*PostExtractFn)(void *ctx, bool ok);
    void SetPostExtract(PostExtractFn fn, void *ctx) { mPostFn = fn; mPostCtx = ctx; }
    // Opaque per-extractor state of shim/Frame_hip.cc (its matcher / frame-ops handles for frames built with this extractor as the left
    // one); freed with the extractor.
    void *mpFrameAssist;
    void (*mpFrameAssistFree)(void *);

    // Error channel (the reference has none: include/ORBextractor.h:92-161).  A failed call leaves `keypoints` EMPTY and `descriptors`
    // released - never the previous frame's data -, increments ErrorCount() and keeps the message; a host program polls these instead of
    // scraping stderr.  A device that could not be opened at construction is not retried on every frame (Dead()).
    int ErrorCount() const { return mnErrors; }
    const std::string &LastError() const { return mLastError; }
    bool Dead() const { return mbDead; }
    static bool sbThrowOnError;     // default: false, or true when ORBX_SHIM_FATAL=1 is set in the environment

    // Device used by extractors constructed afterwards (default 0).
    static void SetDevice(int device);
    // liborbx handle holding the device-resident results and pyramid of the last frame
    // (consumed by the HIP body of Frame::ComputeStereoMatches, shim/Frame_hip.cc).
    orbx_extractor *Handle() const { return mpHandle; }

private:
    ORBextractor(const ORBextractor &);
    ORBextractor &operator=(const ORBextractor &);
    bool EnsureHandle(int width, int height);
    bool Fail(const char *what);

    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel;
    orbx_extractor *mpHandle;
    int mMaxW, mMaxH, mLastW, mLastH;
    PostExtractFn mPostFn;
    void *mPostCtx;
    int mnErrors;
    bool mbDead;
    std::string mLastError;
};

} // namespace ORB_SLAM2

#endif

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

    // Host copy of the image pyramid of the last frame: the reference's public `std::vector<cv::Mat> mvImagePyramid` (include/ORBextractor.h:161),
    // whose only reader is Frame::ComputeStereoMatches (src/Frame.cc:1044, 1248, 1272, 1281: `mvImagePyramid[octave]`).  Here the member is a
    // small vector-like class that fills itself on FIRST ACCESS after a call (operator[] / Levels()): every level becomes a freshly allocated,
    // OWNING cv::Mat - as the reference's ComputePyramid leaves them (src/ORBextractor.cc:1687-1689): a level a caller keeps across calls stays
    // valid and unchanged, and nothing points into the handle's memory.  Where the bytes come from:
    //   mbKeepHostPyramid (TRUE unless shim/Frame_hip.cc is linked into the same binary - its ComputeStereoMatches reads the DEVICE pyramid) and
    //   the pyramid was read after the previous call: the levels came back with the results of the same launch set (pinned memory of the handle,
    //   no second transfer, no second wait) and the first access is one copy out of it;
    //   otherwise (never read so far, or mbKeepHostPyramid false): nothing crosses PCIe per frame, and a first access downloads the frame's
    //   pyramid then (one extra transfer, ~0.1 ms), after which the following calls bring it along again.
    // A failed call leaves every level empty.  DownloadImagePyramid() forces the (owning) copies now.
    // mbViewHostPyramid (default FALSE; opt-in for a caller that reads the levels before its next call and keeps none of them): the levels
    // are VIEWS of the handle's pinned memory instead of copies (0.64 MB and ~45 us less per 640x480 frame) - overwritten by the next call,
    // emptied by the extractor before its handle is rebuilt or destroyed; a header copy a caller made of such a level is the caller's risk.
    // It binds like the member it replaces: `std::vector<cv::Mat> &v = ex.mvImagePyramid;` (conversion, filling first), range-for / begin() / end(),
    // at(), front() / back(), size() / empty() / resize() / clear(); every accessor that hands out a level fills first.
    class ImagePyramid
    {
    public:
        typedef std::vector<cv::Mat>::iterator iterator;
        typedef std::vector<cv::Mat>::const_iterator const_iterator;
        typedef cv::Mat value_type;
        ImagePyramid() : mpOwner(0) {}
        cv::Mat &operator[](size_t level) { Fill(); return mv[level]; }
        const cv::Mat &operator[](size_t level) const { const_cast<ImagePyramid *>(this)->Fill(); return mv[level]; }
        cv::Mat &at(size_t level) { Fill(); return mv.at(level); }
        const cv::Mat &at(size_t level) const { const_cast<ImagePyramid *>(this)->Fill(); return mv.at(level); }
        cv::Mat &front() { Fill(); return mv.front(); }
        cv::Mat &back() { Fill(); return mv.back(); }
        const cv::Mat &front() const { const_cast<ImagePyramid *>(this)->Fill(); return mv.front(); }
        const cv::Mat &back() const { const_cast<ImagePyramid *>(this)->Fill(); return mv.back(); }
        iterator begin() { Fill(); return mv.begin(); }
        iterator end() { Fill(); return mv.end(); }
        const_iterator begin() const { const_cast<ImagePyramid *>(this)->Fill(); return mv.begin(); }
        const_iterator end() const { const_cast<ImagePyramid *>(this)->Fill(); return mv.end(); }
        size_t size() const { return mv.size(); }
        bool empty() const { return mv.empty(); }
        void resize(size_t n) { mv.resize(n); }
        void clear() { if (mpOwner) mpOwner->DropImagePyramid(); for (size_t i = 0; i < mv.size(); i++) mv[i] = cv::Mat(); }      // (the levels stay: nlevels is the extractor's)
        operator std::vector<cv::Mat> &() { Fill(); return mv; }
        operator const std::vector<cv::Mat> &() const { const_cast<ImagePyramid *>(this)->Fill(); return mv; }
        std::vector<cv::Mat> &Levels() { Fill(); return mv; }
    private:
        friend class ORBextractor;
        void Fill() { if (mpOwner) mpOwner->FillImagePyramid(); }
        std::vector<cv::Mat> mv;
        ORBextractor *mpOwner;
    };
    ImagePyramid mvImagePyramid;
    bool mbKeepHostPyramid;
    bool mbViewHostPyramid;
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
    void FillImagePyramid();          // first access of mvImagePyramid after a call
    void DropImagePyramid();          // every level empty, nothing pending (failed call, handle about to go)
    enum { PYR_NONE = 0, PYR_IN_PINNED = 1, PYR_ON_DEVICE = 2, PYR_OWNED = 3, PYR_VIEWS = 4 };
    int mPyrState;                    // where the last frame's pyramid is
    bool mbPyrRead;                   // ... and whether it was read since the last call (the next call then brings the pyramid along)
    const unsigned char *mPyrLevel[12];
    int mPyrW[12], mPyrH[12], mPyrStride[12], mPyrLevels;

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

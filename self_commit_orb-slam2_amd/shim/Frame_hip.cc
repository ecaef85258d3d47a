// shim/Frame_hip.cc -- HIP body for ORB_SLAM2::Frame::ComputeStereoMatches.
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

static unsigned long gStereoCalls = 0;
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

void Frame::ComputeStereoMatches()
{
    __atomic_add_fetch(&gStereoCalls, 1, __ATOMIC_RELAXED);
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
    const int32_t zero = 0;
    if (orbx_compute_stereo_matches_device(tStereo.h, hl, hr, &zero, &zero, 1, mbf, mb) != ORBX_OK ||
        orbx_stereo_download(tStereo.h, 1, &mvuRight[0], &mvDepth[0], N) != ORBX_OK)
        throw std::runtime_error(std::string("Frame::ComputeStereoMatches (orbx): ") + orbx_last_error());
}

}  // namespace ORB_SLAM2

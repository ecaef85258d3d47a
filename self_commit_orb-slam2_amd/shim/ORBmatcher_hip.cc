// shim/ORBmatcher_hip.cc -- HIP bodies for the Hamming paths of ORB_SLAM2::ORBmatcher.
//
// Compiled against the REFERENCE's own include/ORBmatcher.h (no header change): this file
// replaces the bodies of
//     int ORBmatcher::SearchByBoW(KeyFrame*, Frame&, std::vector<MapPoint*>&)       src/ORBmatcher.cc:230-382
//     int ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, std::vector<MapPoint*>&)    src/ORBmatcher.cc:656-799
//     int ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)            src/ORBmatcher.cc:1913-1933
//     int ORBmatcher::SearchByProjection(Frame&, const std::vector<MapPoint*>&, float)   src/ORBmatcher.cc:70-175
//     int ORBmatcher::SearchByProjection(Frame&, const Frame&, float, bool)              src/ORBmatcher.cc:1569-1728
//     int ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat, const std::vector<MapPoint*>&, std::vector<MapPoint*>&, int)   src/ORBmatcher.cc:388-513
//     int ORBmatcher::SearchByProjection(Frame&, KeyFrame*, const std::set<MapPoint*>&, float, int)                         src/ORBmatcher.cc:1731-1864
//     int ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat, std::vector<std::pair<size_t,size_t> >&, bool)   src/ORBmatcher.cc:810-1017
//     int ORBmatcher::Fuse(KeyFrame*, const std::vector<MapPoint*>&, float)                                                src/ORBmatcher.cc:1020-1177
//     int ORBmatcher::Fuse(KeyFrame*, cv::Mat, const std::vector<MapPoint*>&, float, std::vector<MapPoint*>&)              src/ORBmatcher.cc:1179-1312
//     int ORBmatcher::SearchForInitialization(Frame&, Frame&, std::vector<cv::Point2f>&, std::vector<int>&, int)               src/ORBmatcher.cc:515-654
//     int ORBmatcher::SearchBySim3(KeyFrame*, KeyFrame*, std::vector<MapPoint*>&, const float&, const cv::Mat&, const cv::Mat&, float)   src/ORBmatcher.cc:1314-1523
// A maintainer deletes those three bodies from src/ORBmatcher.cc and adds this file to the
// source list; the test build keeps src/ORBmatcher.cc untouched and weakens
// the three symbols in its object file instead (oracle/Makefile, target liborbslam_hip.so).
// Everything the reference reads from the object graph is marshalled into the flat arrays of
// include/orbx.h; the order-dependent greedy assignment, the ratio test and the rotation
// histogram run on the device and are index-exact (tests/test_dropin_slam.py).
#include <string.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "ORBmatcher.h"
#include "SearchLocalPoints.h"
#include "MapPointAccess.h"
#include "orbx.h"
#include "shim_error.h"

// number of SearchByBoW calls served by this file (lets the drop-in test prove that the HIP
// bodies, not the reference's, were linked)
static unsigned long gSearchByProjectionCalls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_search_by_projection_calls(void) { return gSearchByProjectionCalls; }
static unsigned long gInitCalls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_search_for_initialization_calls(void) { return gInitCalls; }
static unsigned long gSim3Calls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_search_by_sim3_calls(void) { return gSim3Calls; }
static unsigned long gFuseCalls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_fuse_calls(void) { return gFuseCalls; }
static unsigned long gTriangulationCalls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_search_for_triangulation_calls(void) { return gTriangulationCalls; }
static unsigned long gLocalPointsCalls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_search_local_points_calls(void) { return gLocalPointsCalls; }
static unsigned long gSearchByBoWCalls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_search_by_bow_calls(void) { return gSearchByBoWCalls; }

namespace ORB_SLAM2
{

// ---------------------------------------------------------------------------------------------
// With every search function below defined here, src/ORBmatcher.cc leaves the build entirely; what remains
// of it are the constants, the constructor and three small helpers (src/ORBmatcher.cc:49-58, 178-227, 1866-1908).
// ---------------------------------------------------------------------------------------------
const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

float ORBmatcher::RadiusByViewingCos(const float &viewCos) { return viewCos > 0.998 ? 2.5 : 4.0; }

bool ORBmatcher::CheckDistEpipolarLine(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame *pKF2)
{
    // the epipolar line of kp1 in the second image, l = x1' F12 = [a b c], and the squared distance of kp2 to it
    const float a = kp1.pt.x * F12.at<float>(0, 0) + kp1.pt.y * F12.at<float>(1, 0) + F12.at<float>(2, 0);
    const float b = kp1.pt.x * F12.at<float>(0, 1) + kp1.pt.y * F12.at<float>(1, 1) + F12.at<float>(2, 1);
    const float c = kp1.pt.x * F12.at<float>(0, 2) + kp1.pt.y * F12.at<float>(1, 2) + F12.at<float>(2, 2);
    const float num = a * kp2.pt.x + b * kp2.pt.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave];
}

void ORBmatcher::ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3)
{
    int best[3] = {0, 0, 0};
    int *ind[3] = {&ind1, &ind2, &ind3};
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        int slot = s > best[0] ? 0 : (s > best[1] ? 1 : (s > best[2] ? 2 : 3));
        for (int k = 2; k > slot; k--) { best[k] = best[k - 1]; *ind[k] = *ind[k - 1]; }
        if (slot < 3) { best[slot] = s; *ind[slot] = i; }
    }
    if (best[1] < 0.1f * (float)best[0]) { ind2 = -1; ind3 = -1; }
    else if (best[2] < 0.1f * (float)best[0]) ind3 = -1;
}


static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint must be the 28-byte layout of orbx_keypoint");

namespace
{
// One matcher handle per calling thread: SearchByBoW is called from the Tracking thread
// (src/Tracking.cc:1195, 2073) and from LoopClosing (src/LoopClosing.cc:375) concurrently,
// and a handle is not re-entrant.
struct ThreadMatcher {
    orbx_matcher *h;
    int cap;
    ThreadMatcher() : h(0), cap(0) {}
    ~ThreadMatcher() { if (h) orbx_matcher_destroy(h); }
};
thread_local ThreadMatcher tMatcher;

orbx_matcher *Matcher(int need)
{
    if (tMatcher.h && need <= tMatcher.cap) return tMatcher.h;
    if (tMatcher.h) { orbx_matcher_destroy(tMatcher.h); tMatcher.h = 0; }
    int cap = 4096;
    while (cap < need) cap *= 2;
    if (orbx_matcher_create(orbx_shim::Device(), cap, 1, &tMatcher.h) != ORBX_OK) { tMatcher.h = 0; tMatcher.cap = 0; orbx_shim::Fail("ORBmatcher"); return 0; }   // (the call that follows fails on the NULL handle)
    tMatcher.cap = cap;
    return tMatcher.h;
}

// DBoW2::FeatureVector (node id -> feature indices) as one node id per feature.  Features the
// vocabulary did not file (word weight 0) get -1: the C ABI never matches a negative node id.
void FlatGroups(const DBoW2::FeatureVector &fv, int N, std::vector<int32_t> &g)
{
    g.assign((size_t)(N > 0 ? N : 1), -1);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it)
        for (size_t k = 0; k < it->second.size(); k++)
            if ((int)it->second[k] < N) g[it->second[k]] = (int32_t)it->first;
}

void ValidMask(const std::vector<MapPoint *> &vp, std::vector<uint8_t> &valid)
{
    valid.assign(vp.size() ? vp.size() : 1, 0);
    for (size_t i = 0; i < vp.size(); i++) valid[i] = (vp[i] && !vp[i]->isBad()) ? 1 : 0;   // src/ORBmatcher.cc:268-274, 714-721
}
}  // namespace

int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
{
    return orbx_descriptor_distance(a.ptr<unsigned char>(), b.ptr<unsigned char>());
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches)
{
    __atomic_add_fetch(&gSearchByBoWCalls, 1, __ATOMIC_RELAXED);
    orbx_shim::Mark("SearchByBoW enters");
    const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();                 // :232 (locks inside)
    vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));         // :236
    const int NA = (int)vpMapPointsKF.size(), NB = F.N;
    if (NA == 0 || NB == 0) return 0;
    std::vector<int32_t> gA, gB;
    FlatGroups(pKF->mFeatVec, NA, gA);
    FlatGroups(F.mFeatVec, NB, gB);
    std::vector<uint8_t> validA;
    ValidMask(vpMapPointsKF, validA);
    orbx_feature_set a = {(const orbx_keypoint *)&pKF->mvKeysUn[0], pKF->mDescriptors.data, &NA, &gA[0], &validA[0], NA, 1};   // kp angle: :318
    orbx_feature_set b = {(const orbx_keypoint *)&F.mvKeys[0], F.mDescriptors.data, &NB, &gB[0], NULL, NB, 1};                 // :325
    orbx_bow_params prm = {mfNNratio, mbCheckOrientation ? 1 : 0, 0};
    std::vector<int32_t> match((size_t)NB);
    int32_t nmatches = 0;
    orbx_shim::Mark("SearchByBoW marshalled");
    if (orbx_search_by_bow(Matcher(NA > NB ? NA : NB), &a, &b, &prm, &match[0], &nmatches) != ORBX_OK)
        { orbx_shim::Fail("ORBmatcher::SearchByBoW"); return 0; }
    orbx_shim::Mark("SearchByBoW device call returned");
    for (int j = 0; j < NB; j++)
        if (match[(size_t)j] >= 0) vpMapPointMatches[(size_t)j] = vpMapPointsKF[(size_t)match[(size_t)j]];                   // :314
    orbx_shim::Mark("SearchByBoW returns");
    return nmatches;
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12)
{
    __atomic_add_fetch(&gSearchByBoWCalls, 1, __ATOMIC_RELAXED);
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();                 // :661
    const std::vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();                 // :667
    vpMatches12 = std::vector<MapPoint *>(vpMapPoints1.size(), static_cast<MapPoint *>(NULL));   // :672
    const int NA = (int)vpMapPoints1.size(), NB = (int)vpMapPoints2.size();
    if (NA == 0 || NB == 0) return 0;
    std::vector<int32_t> gA, gB;
    FlatGroups(pKF1->mFeatVec, NA, gA);
    FlatGroups(pKF2->mFeatVec, NB, gB);
    std::vector<uint8_t> validA, validB;
    ValidMask(vpMapPoints1, validA);
    ValidMask(vpMapPoints2, validB);
    orbx_feature_set a = {(const orbx_keypoint *)&pKF1->mvKeysUn[0], pKF1->mDescriptors.data, &NA, &gA[0], &validA[0], NA, 1};   // :750
    orbx_feature_set b = {(const orbx_keypoint *)&pKF2->mvKeysUn[0], pKF2->mDescriptors.data, &NB, &gB[0], &validB[0], NB, 1};
    orbx_bow_params prm = {mfNNratio, mbCheckOrientation ? 1 : 0, 1};
    std::vector<int32_t> match((size_t)NA);
    int32_t nmatches = 0;
    if (orbx_search_by_bow(Matcher(NA > NB ? NA : NB), &a, &b, &prm, &match[0], &nmatches) != ORBX_OK)
        { orbx_shim::Fail("ORBmatcher::SearchByBoW"); return 0; }
    for (int i = 0; i < NA; i++)
        if (match[(size_t)i] >= 0) vpMatches12[(size_t)i] = vpMapPoints2[(size_t)match[(size_t)i]];                           // :745
    return nmatches;
}

// LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:332): features of the two KeyFrames that hold
// no MapPoint yet, matched inside their vocabulary nodes under the epipolar constraint of F12.
int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                                       const bool bOnlyStereo)
{
    __atomic_add_fetch(&gTriangulationCalls, 1, __ATOMIC_RELAXED);
    // the epipole of KF1 in KF2, :817-826 (cv::Mat arithmetic of the reference, unchanged)
    cv::Mat Cw = pKF1->GetCameraCenter();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2.at<float>(2);
    const float epipole[2] = {pKF2->fx * C2.at<float>(0) * invz + pKF2->cx, pKF2->fy * C2.at<float>(1) * invz + pKF2->cy};
    vMatchedPairs.clear();
    const int NA = pKF1->N, NB = pKF2->N;
    if (NA == 0 || NB == 0) return 0;
    std::vector<int32_t> gA, gB;
    FlatGroups(pKF1->mFeatVec, NA, gA);
    FlatGroups(pKF2->mFeatVec, NB, gB);
    std::vector<uint8_t> okA((size_t)NA), okB((size_t)NB), stA((size_t)NA), stB((size_t)NB);
    for (int i = 0; i < NA; i++) {
        stA[(size_t)i] = pKF1->mvuRight[(size_t)i] >= 0 ? 1 : 0;                                                    // :850
        okA[(size_t)i] = (!pKF1->GetMapPoint((size_t)i) && (!bOnlyStereo || stA[(size_t)i])) ? 1 : 0;              // :845-855
    }
    for (int i = 0; i < NB; i++) {
        stB[(size_t)i] = pKF2->mvuRight[(size_t)i] >= 0 ? 1 : 0;                                                    // :872
        okB[(size_t)i] = (!pKF2->GetMapPoint((size_t)i) && (!bOnlyStereo || stB[(size_t)i])) ? 1 : 0;              // :867-876
    }
    float f12[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) f12[3 * r + c] = F12.at<float>(r, c);
    orbx_feature_set a = {(const orbx_keypoint *)&pKF1->mvKeysUn[0], pKF1->mDescriptors.data, &NA, &gA[0], &okA[0], NA, 1};
    orbx_feature_set b = {(const orbx_keypoint *)&pKF2->mvKeysUn[0], pKF2->mDescriptors.data, &NB, &gB[0], &okB[0], NB, 1};
    orbx_triangulation_params prm = {f12, epipole, &stA[0], &stB[0], &pKF2->mvScaleFactors[0], &pKF2->mvLevelSigma2[0],
                                     (int)pKF2->mvScaleFactors.size(), mbCheckOrientation ? 1 : 0};
    std::vector<int32_t> match((size_t)NA);
    int32_t nmatches = 0;
    if (orbx_search_for_triangulation(Matcher(NA > NB ? NA : NB), &a, &b, &prm, &match[0], &nmatches) != ORBX_OK)
        { orbx_shim::Fail("ORBmatcher::SearchForTriangulation"); return 0; }
    vMatchedPairs.reserve((size_t)nmatches);                                                                          // :1005-1014
    for (int i = 0; i < NA; i++)
        if (match[(size_t)i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)match[(size_t)i]));
    return nmatches;
}

// ---------------------------------------------------------------------------------------------
// Fuse, both overloads.  Per map point the reference (1) projects it and applies the visibility gates,
// (2) searches the KeyFrame's features around the projection, (3) rewires pointers.  (1) is a handful
// of cv::Mat expressions per point and PredictScale (a libm log): kept verbatim on the host; (2) is the
// data-parallel part: one orbx_fuse_search call for the whole list; (3) stays the reference's code,
// run in list order on the results.  A point that is bad or already in the KeyFrame when the call
// starts is still so at its turn (Replace only ever removes points from play), so filtering (1)+(2)
// up front does not change what (3) sees.
// ---------------------------------------------------------------------------------------------
namespace
{
struct FuseArrays {
    std::vector<float> u, v, ur, radius;
    std::vector<int32_t> level, bestIdx, bestDist;
    std::vector<uint8_t> active, desc;
    explicit FuseArrays(size_t n) : u(n), v(n), ur(n), radius(n), level(n), bestIdx(n, -1), bestDist(n, 256), active(n, 0), desc(n * 32, 0) {}
};

void FuseSearch(KeyFrame *pKF, FuseArrays &A, int n, int chi2Gate)
{
    if (n == 0 || pKF->N == 0) return;
    const int N = pKF->N;
    orbx_projection_frame kf = {(const orbx_keypoint *)&pKF->mvKeysUn[0], pKF->mDescriptors.data, &pKF->mvuRight[0], 0, &N, N, 1,
                                Frame::mnMinX, Frame::mnMinY, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv};   // what filed mGrid
    orbx_fuse_points pt = {&A.u[0], &A.v[0], &A.ur[0], &A.level[0], &A.radius[0], &A.active[0], &A.desc[0], &n, n,
                           (float)pKF->mnMinX, (float)pKF->mnMinY};                                                         // what the window uses
    if (orbx_fuse_search(Matcher(N > n ? N : n), &kf, &pt, &pKF->mvInvLevelSigma2[0], (int)pKF->mvInvLevelSigma2.size(), chi2Gate, &A.bestIdx[0],
                         &A.bestDist[0]) != ORBX_OK)
        { orbx_shim::Fail("ORBmatcher::Fuse"); return; }
}
}  // namespace

int ORBmatcher::Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th)
{
    __atomic_add_fetch(&gFuseCalls, 1, __ATOMIC_RELAXED);
    cv::Mat Rcw = pKF->GetRotation();
    cv::Mat tcw = pKF->GetTranslation();
    const float &fx = pKF->fx;
    const float &fy = pKF->fy;
    const float &cx = pKF->cx;
    const float &cy = pKF->cy;
    const float &bf = pKF->mbf;
    cv::Mat Ow = pKF->GetCameraCenter();
    const int nMPs = (int)vpMapPoints.size();
    FuseArrays A((size_t)nMPs);
    for (int i = 0; i < nMPs; i++) {                                            // step 1, :1032-1093
        MapPoint *pMP = vpMapPoints[(size_t)i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0f) continue;
        const float invz = 1 / p3Dc.at<float>(2);
        const float x = p3Dc.at<float>(0) * invz;
        const float y = p3Dc.at<float>(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float ur = u - bf * invz;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow;
        const float dist3D = cv::norm(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist3D) continue;
        int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        const cv::Mat dMP = pMP->GetDescriptor();
        A.u[(size_t)i] = u; A.v[(size_t)i] = v; A.ur[(size_t)i] = ur; A.level[(size_t)i] = nPredictedLevel; A.radius[(size_t)i] = radius;
        A.active[(size_t)i] = 1;
        memcpy(&A.desc[32 * (size_t)i], dMP.ptr<unsigned char>(), 32);
    }
    FuseSearch(pKF, A, nMPs, 1);                                                // steps 2-3, :1093-1146
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {                                            // :1148-1174, in list order
        MapPoint *pMP = vpMapPoints[(size_t)i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        if (!A.active[(size_t)i]) continue;
        const int bestDist = A.bestDist[(size_t)i], bestIdx = A.bestIdx[(size_t)i];
        if (bestDist <= TH_LOW) {
            MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
    }
    return nFused;
}

int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint)
{
    __atomic_add_fetch(&gFuseCalls, 1, __ATOMIC_RELAXED);
    const float &fx = pKF->fx;
    const float &fy = pKF->fy;
    const float &cx = pKF->cx;
    const float &cy = pKF->cy;
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();
    const int nPoints = (int)vpPoints.size();
    FuseArrays A((size_t)nPoints);
    for (int iMP = 0; iMP < nPoints; iMP++) {                                   // :1205-1258
        MapPoint *pMP = vpPoints[(size_t)iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0f) continue;
        const float invz = 1.0 / p3Dc.at<float>(2);
        const float x = p3Dc.at<float>(0) * invz;
        const float y = p3Dc.at<float>(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow;
        const float dist3D = cv::norm(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist3D) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        const cv::Mat dMP = pMP->GetDescriptor();
        A.u[(size_t)iMP] = u; A.v[(size_t)iMP] = v; A.level[(size_t)iMP] = nPredictedLevel; A.radius[(size_t)iMP] = radius;
        A.active[(size_t)iMP] = 1;
        memcpy(&A.desc[32 * (size_t)iMP], dMP.ptr<unsigned char>(), 32);
    }
    FuseSearch(pKF, A, nPoints, 0);                                             // :1258-1276
    int nFused = 0;
    for (int iMP = 0; iMP < nPoints; iMP++) {                                   // :1278-1306
        if (!A.active[(size_t)iMP]) continue;
        MapPoint *pMP = vpPoints[(size_t)iMP];
        const int bestDist = A.bestDist[(size_t)iMP], bestIdx = A.bestIdx[(size_t)iMP];
        if (bestDist <= TH_LOW) {
            MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[(size_t)iMP] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
    }
    return nFused;
}

// LoopClosing::ComputeSim3 (src/LoopClosing.cc): the MapPoints of each KeyFrame are carried into the other
// one with the Sim3 estimate and searched there; a match must be mutual.  The two searches are the Fuse
// search without the chi-square gate (level gate, first minimum in GetFeaturesInArea order), accepted up to
// TH_HIGH; the per-point preparation is the reference's own host code.
namespace
{
// one direction of :1352-1427 / :1430-1504: points of pKFa searched in pKFb; Rcw/tcw of pKFa, (sR, t) into pKFb's camera
void Sim3Direction(KeyFrame *pKFb, const std::vector<MapPoint *> &vpMapPointsA, const std::vector<bool> &vbAlreadyMatchedA, const cv::Mat &Raw,
                   const cv::Mat &taw, const cv::Mat &sRba, const cv::Mat &tba, float fx, float fy, float cx, float cy, float th, int thDist,
                   std::vector<int> &vnMatchA)
{
    const int NA = (int)vpMapPointsA.size();
    FuseArrays A((size_t)NA);
    for (int i = 0; i < NA; i++) {
        MapPoint *pMP = vpMapPointsA[(size_t)i];
        if (!pMP || vbAlreadyMatchedA[(size_t)i]) continue;
        if (pMP->isBad()) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dca = Raw * p3Dw + taw;
        cv::Mat p3Dcb = sRba * p3Dca + tba;
        if (p3Dcb.at<float>(2) < 0.0) continue;
        const float invz = 1.0 / p3Dcb.at<float>(2);
        const float x = p3Dcb.at<float>(0) * invz;
        const float y = p3Dcb.at<float>(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKFb->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        const float dist3D = cv::norm(p3Dcb);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKFb);
        const float radius = th * pKFb->mvScaleFactors[nPredictedLevel];
        const cv::Mat dMP = pMP->GetDescriptor();
        A.u[(size_t)i] = u; A.v[(size_t)i] = v; A.level[(size_t)i] = nPredictedLevel; A.radius[(size_t)i] = radius;
        A.active[(size_t)i] = 1;
        memcpy(&A.desc[32 * (size_t)i], dMP.ptr<unsigned char>(), 32);
    }
    FuseSearch(pKFb, A, NA, 0);
    for (int i = 0; i < NA; i++)
        if (A.active[(size_t)i] && A.bestDist[(size_t)i] <= thDist) vnMatchA[(size_t)i] = A.bestIdx[(size_t)i];
}
}  // namespace

int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12,
                             const float th)
{
    __atomic_add_fetch(&gSim3Calls, 1, __ATOMIC_RELAXED);
    const float &fx = pKF1->fx;
    const float &fy = pKF1->fy;
    const float &cx = pKF1->cx;
    const float &cy = pKF1->cy;
    cv::Mat R1w = pKF1->GetRotation();
    cv::Mat t1w = pKF1->GetTranslation();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat sR12 = s12 * R12;
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -sR21 * t12;
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size();
    const std::vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1((size_t)N1, false);
    std::vector<bool> vbAlreadyMatched2((size_t)N2, false);
    for (int i = 0; i < N1; i++) {                                              // :1337-1348
        MapPoint *pMP = vpMatches12[(size_t)i];
        if (pMP) {
            vbAlreadyMatched1[(size_t)i] = true;
            int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[(size_t)idx2] = true;
        }
    }
    std::vector<int> vnMatch1((size_t)N1, -1);
    std::vector<int> vnMatch2((size_t)N2, -1);
    Sim3Direction(pKF2, vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, fx, fy, cx, cy, th, TH_HIGH, vnMatch1);   // :1352-1427
    Sim3Direction(pKF1, vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12, fx, fy, cx, cy, th, TH_HIGH, vnMatch2);   // :1430-1504
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {                                           // :1507-1521
        int idx2 = vnMatch1[(size_t)i1];
        if (idx2 >= 0) {
            int idx1 = vnMatch2[(size_t)idx2];
            if (idx1 == i1) {
                vpMatches12[(size_t)i1] = vpMapPoints2[(size_t)idx2];
                nFound++;
            }
        }
    }
    return nFound;
}

// ---------------------------------------------------------------------------------------------
// The two remaining projection searches: loop closing (LoopClosing::ComputeSim3 / CorrectLoop) and
// relocalisation (Tracking::Relocalization).  Same split as Fuse: the reference's per-point preparation on the
// host, one orbx_area_search_greedy call for the order-dependent search, the reference's bookkeeping after it.
// ---------------------------------------------------------------------------------------------
namespace
{
struct AreaArrays {
    std::vector<float> u, v, radius;
    std::vector<int32_t> lo, hi, assigned, dist;
    std::vector<uint8_t> active, desc;
    explicit AreaArrays(size_t n) : u(n), v(n), radius(n), lo(n), hi(n), assigned(n, -1), dist(n, 256), active(n, 0), desc(n * 32, 0) {}
};
}  // namespace

int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th)
{
    __atomic_add_fetch(&gSearchByProjectionCalls, 1, __ATOMIC_RELAXED);
    const float &fx = pKF->fx;
    const float &fy = pKF->fy;
    const float &cx = pKF->cx;
    const float &cy = pKF->cy;
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint *>(NULL));
    const int nPoints = (int)vpPoints.size(), N = pKF->N;
    if (nPoints == 0 || N == 0) return 0;
    AreaArrays A((size_t)nPoints);
    for (int iMP = 0; iMP < nPoints; iMP++) {                                   // :405-452
        MapPoint *pMP = vpPoints[(size_t)iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0) continue;
        const float invz = 1 / p3Dc.at<float>(2);
        const float x = p3Dc.at<float>(0) * invz;
        const float y = p3Dc.at<float>(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow;
        const float dist = cv::norm(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist) continue;
        int nPredictedLevel = pMP->PredictScale(dist, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        const cv::Mat dMP = pMP->GetDescriptor();
        A.u[(size_t)iMP] = u; A.v[(size_t)iMP] = v; A.radius[(size_t)iMP] = radius;
        A.lo[(size_t)iMP] = nPredictedLevel - 1; A.hi[(size_t)iMP] = nPredictedLevel;   // :469-470
        A.active[(size_t)iMP] = 1;
        memcpy(&A.desc[32 * (size_t)iMP], dMP.ptr<unsigned char>(), 32);
    }
    std::vector<uint8_t> blocked((size_t)N);
    for (int i = 0; i < N; i++) blocked[(size_t)i] = vpMatched[(size_t)i] ? 1 : 0;                                              // :465-466
    orbx_projection_frame kf = {(const orbx_keypoint *)&pKF->mvKeysUn[0], pKF->mDescriptors.data, 0, &blocked[0], &N, N, 1,
                                Frame::mnMinX, Frame::mnMinY, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv};
    orbx_area_queries q = {&A.u[0], &A.v[0], &A.radius[0], &A.lo[0], &A.hi[0], &A.active[0], &A.desc[0], &nPoints, nPoints,
                           (float)pKF->mnMinX, (float)pKF->mnMinY};
    int32_t nm = 0;
    if (orbx_area_search_greedy(Matcher(N > nPoints ? N : nPoints), &kf, &q, TH_LOW, &A.assigned[0], &A.dist[0], &nm) != ORBX_OK)
        { orbx_shim::Fail("ORBmatcher::SearchByProjection"); return 0; }
    int nmatches = 0;
    for (int iMP = 0; iMP < nPoints; iMP++)                                                                                     // :505-509
        if (A.assigned[(size_t)iMP] >= 0) { vpMatched[(size_t)A.assigned[(size_t)iMP]] = vpPoints[(size_t)iMP]; nmatches++; }
    return nmatches;
}

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist)
{
    __atomic_add_fetch(&gSearchByProjectionCalls, 1, __ATOMIC_RELAXED);
    int nmatches = 0;
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    std::vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
    const float factor = HISTO_LENGTH / 360.0f;
    const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
    const int nPoints = (int)vpMPs.size(), N = CurrentFrame.N;
    if (nPoints == 0 || N == 0) return 0;
    AreaArrays A((size_t)nPoints);
    for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {                    // :1747-1790
        MapPoint *pMP = vpMPs[i];
        if (pMP) {
            if (!pMP->isBad() && !sAlreadyFound.count(pMP)) {
                cv::Mat x3Dw = pMP->GetWorldPos();
                cv::Mat x3Dc = Rcw * x3Dw + tcw;
                const float xc = x3Dc.at<float>(0);
                const float yc = x3Dc.at<float>(1);
                const float invzc = 1.0 / x3Dc.at<float>(2);
                const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
                const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
                if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
                if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
                cv::Mat PO = x3Dw - Ow;
                float dist3D = cv::norm(PO);
                const float maxDistance = pMP->GetMaxDistanceInvariance();
                const float minDistance = pMP->GetMinDistanceInvariance();
                if (dist3D < minDistance || dist3D > maxDistance) continue;
                int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
                const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
                const cv::Mat dMP = pMP->GetDescriptor();
                A.u[i] = u; A.v[i] = v; A.radius[i] = radius;
                A.lo[i] = nPredictedLevel - 1; A.hi[i] = nPredictedLevel + 1;   // :1792
                A.active[i] = 1;
                memcpy(&A.desc[32 * i], dMP.ptr<unsigned char>(), 32);
            }
        }
    }
    std::vector<uint8_t> blocked((size_t)N);
    for (int i = 0; i < N; i++) blocked[(size_t)i] = CurrentFrame.mvpMapPoints[(size_t)i] ? 1 : 0;                              // :1805-1806
    orbx_projection_frame fr = {(const orbx_keypoint *)&CurrentFrame.mvKeysUn[0], CurrentFrame.mDescriptors.data, 0, &blocked[0], &N, N, 1,
                                Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
    orbx_area_queries q = {&A.u[0], &A.v[0], &A.radius[0], &A.lo[0], &A.hi[0], &A.active[0], &A.desc[0], &nPoints, nPoints, Frame::mnMinX, Frame::mnMinY};
    int32_t nm = 0;
    if (orbx_area_search_greedy(Matcher(N > nPoints ? N : nPoints), &fr, &q, ORBdist, &A.assigned[0], &A.dist[0], &nm) != ORBX_OK)
        { orbx_shim::Fail("ORBmatcher::SearchByProjection"); return 0; }
    for (int i = 0; i < nPoints; i++) {                                         // :1819-1840, in the reference's order
        const int bestIdx2 = A.assigned[(size_t)i];
        if (bestIdx2 < 0) continue;
        CurrentFrame.mvpMapPoints[(size_t)bestIdx2] = vpMPs[(size_t)i];
        nmatches++;
        if (mbCheckOrientation) {
            float rot = pKF->mvKeysUn[(size_t)i].angle - CurrentFrame.mvKeysUn[(size_t)bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back(bestIdx2);
        }
    }
    if (mbCheckOrientation) {                                                   // :1844-1861
        int ind1 = -1;
        int ind2 = -1;
        int ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    CurrentFrame.mvpMapPoints[(size_t)rotHist[i][j]] = NULL;
                    nmatches--;
                }
            }
        }
    }
    return nmatches;
}

// Tracking::MonocularInitialization (src/Tracking.cc:944): the whole order-dependent search runs on the device.
int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize)
{
    __atomic_add_fetch(&gInitCalls, 1, __ATOMIC_RELAXED);
    const int N1 = (int)F1.mvKeysUn.size(), N2 = (int)F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>((size_t)N1, -1);                              // :518
    if (N1 == 0 || N2 == 0) return 0;
    std::vector<float> prev((size_t)N1 * 2);
    for (int i = 0; i < N1; i++) { prev[2 * (size_t)i] = vbPrevMatched[(size_t)i].x; prev[2 * (size_t)i + 1] = vbPrevMatched[(size_t)i].y; }
    orbx_feature_set f1 = {(const orbx_keypoint *)&F1.mvKeysUn[0], F1.mDescriptors.data, &N1, 0, 0, N1, 1};
    orbx_projection_frame f2 = {(const orbx_keypoint *)&F2.mvKeysUn[0], F2.mDescriptors.data, 0, 0, &N2, N2, 1,
                                Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
    std::vector<int32_t> match((size_t)N1);
    int32_t nmatches = 0;
    if (orbx_search_for_initialization(Matcher(N1 > N2 ? N1 : N2), &f1, &f2, &prev[0], windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, &match[0], &nmatches) != ORBX_OK)
        { orbx_shim::Fail("ORBmatcher::SearchForInitialization"); return 0; }
    for (int i1 = 0; i1 < N1; i1++) vnMatches12[(size_t)i1] = match[(size_t)i1];
    for (size_t i1 = 0, iend1 = vnMatches12.size(); i1 < iend1; i1++)           // :646-650
        if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[(size_t)vnMatches12[i1]].pt;
    return nmatches;
}

// Tracking::SearchLocalPoints (src/Tracking.cc:1616): the MapPoints carry what Frame::isInFrustum
// computed (mTrackProjX/Y/XR, mnTrackScaleLevel, mTrackViewCos, mbTrackInView); the frame side is
// mvKeysUn / mDescriptors / mvuRight, the grid constants and which features already hold a MapPoint
// with observations (:110-112).  The greedy pass (a feature taken by an observed point blocks later
// points) is replayed on the device in the reference's order.
int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th)
{
    __atomic_add_fetch(&gSearchByProjectionCalls, 1, __ATOMIC_RELAXED);
    const int N = F.N, M = (int)vpMapPoints.size();
    if (N == 0 || M == 0) return 0;
    std::vector<uint8_t> occupied((size_t)N), inView((size_t)M), hasObs((size_t)M), mpDesc((size_t)M * 32, 0);
    for (int i = 0; i < N; i++) occupied[(size_t)i] = (F.mvpMapPoints[(size_t)i] && F.mvpMapPoints[(size_t)i]->Observations() > 0) ? 1 : 0;
    std::vector<float> px((size_t)M), py((size_t)M), pxr((size_t)M), vc((size_t)M);
    std::vector<int32_t> lvl((size_t)M);
    for (int i = 0; i < M; i++) {
        MapPoint *pMP = vpMapPoints[(size_t)i];
        inView[(size_t)i] = (pMP->mbTrackInView && !pMP->isBad()) ? 1 : 0;                 // :79-84
        if (!inView[(size_t)i]) continue;
        px[(size_t)i] = pMP->mTrackProjX; py[(size_t)i] = pMP->mTrackProjY; pxr[(size_t)i] = pMP->mTrackProjXR;
        lvl[(size_t)i] = pMP->mnTrackScaleLevel; vc[(size_t)i] = pMP->mTrackViewCos;
        hasObs[(size_t)i] = pMP->Observations() > 0 ? 1 : 0;
        const cv::Mat d = pMP->GetDescriptor();                                            // :106
        memcpy(&mpDesc[32 * (size_t)i], d.ptr<unsigned char>(), 32);
    }
    orbx_projection_frame fr = {(const orbx_keypoint *)&F.mvKeysUn[0], F.mDescriptors.data, &F.mvuRight[0], &occupied[0], &N, N, 1,
                                Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
    orbx_projection_points pt = {&px[0], &py[0], &pxr[0], &lvl[0], &vc[0], &inView[0], &hasObs[0], &mpDesc[0], &M, M};
    std::vector<int32_t> assigned((size_t)N);
    int32_t nmatches = 0;
    if (orbx_search_by_projection(Matcher(N), &fr, &pt, &F.mvScaleFactors[0], (int)F.mvScaleFactors.size(), th, mfNNratio, &assigned[0], &nmatches) != ORBX_OK)
        { orbx_shim::Fail("ORBmatcher::SearchByProjection"); return 0; }
    for (int i = 0; i < N; i++)
        if (assigned[(size_t)i] >= 0) F.mvpMapPoints[(size_t)i] = vpMapPoints[(size_t)assigned[(size_t)i]];               // :165
    return nmatches;
}

// Tracking::TrackWithMotionModel (src/Tracking.cc:1433-1441): the last frame's MapPoints are
// projected with the current pose on the device (float arithmetic in the reference's order).
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
{
    __atomic_add_fetch(&gSearchByProjectionCalls, 1, __ATOMIC_RELAXED);
    const int N = CurrentFrame.N, NL = LastFrame.N;
    if (N == 0 || NL == 0) return 0;
    std::vector<uint8_t> occupied((size_t)N), valid((size_t)NL, 0), hasObs((size_t)NL, 0), lastDesc((size_t)NL * 32, 0);
    for (int i = 0; i < N; i++)
        occupied[(size_t)i] = (CurrentFrame.mvpMapPoints[(size_t)i] && CurrentFrame.mvpMapPoints[(size_t)i]->Observations() > 0) ? 1 : 0;   // :1647-1649
    std::vector<float> pos((size_t)NL * 3, 0.f), angle((size_t)NL);
    std::vector<int32_t> octave((size_t)NL);
    for (int i = 0; i < NL; i++) {
        octave[(size_t)i] = LastFrame.mvKeys[(size_t)i].octave;                            // :1628
        angle[(size_t)i] = LastFrame.mvKeysUn[(size_t)i].angle;                            // :1694
        MapPoint *pMP = LastFrame.mvpMapPoints[(size_t)i];
        if (!pMP || LastFrame.mvbOutlier[(size_t)i]) continue;                             // :1604-1608
        valid[(size_t)i] = 1;
        const cv::Mat x3Dw = pMP->GetWorldPos();
        for (int k = 0; k < 3; k++) pos[3 * (size_t)i + k] = x3Dw.at<float>(k);
        hasObs[(size_t)i] = pMP->Observations() > 0 ? 1 : 0;
        const cv::Mat d = pMP->GetDescriptor();
        memcpy(&lastDesc[32 * (size_t)i], d.ptr<unsigned char>(), 32);
    }
    float tc[16], tl[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) { tc[4 * r + c] = CurrentFrame.mTcw.at<float>(r, c); tl[4 * r + c] = LastFrame.mTcw.at<float>(r, c); }
    orbx_projection_frame fr = {(const orbx_keypoint *)&CurrentFrame.mvKeysUn[0], CurrentFrame.mDescriptors.data, &CurrentFrame.mvuRight[0], &occupied[0],
                                &N, N, 1, Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
    orbx_projection_last ls = {&valid[0], &pos[0], &lastDesc[0], &hasObs[0], &octave[0], &angle[0], &NL, NL, tc, tl,
                               Frame::fx, Frame::fy, Frame::cx, Frame::cy, CurrentFrame.mbf, CurrentFrame.mb, Frame::mnMaxX, Frame::mnMaxY};
    std::vector<int32_t> assigned((size_t)N);
    int32_t nmatches = 0;
    if (orbx_search_by_projection_last(Matcher(N > NL ? N : NL), &fr, &ls, &CurrentFrame.mvScaleFactors[0], (int)CurrentFrame.mvScaleFactors.size(), th,
                                       bMono ? 1 : 0, mbCheckOrientation ? 1 : 0, &assigned[0], &nmatches) != ORBX_OK)
        { orbx_shim::Fail("ORBmatcher::SearchByProjection"); return 0; }
    // the reference writes mvpMapPoints[bestIdx2] = pMP as it goes and NULLs the pruned ones (:1688, :1718): features it never
    // touched keep what they held
    for (int i2 = 0; i2 < N; i2++) {
        if (assigned[(size_t)i2] >= 0) CurrentFrame.mvpMapPoints[(size_t)i2] = LastFrame.mvpMapPoints[(size_t)assigned[(size_t)i2]];
        else if (assigned[(size_t)i2] == -2) CurrentFrame.mvpMapPoints[(size_t)i2] = static_cast<MapPoint *>(NULL);
    }
    return nmatches;
}

// ---------------------------------------------------------------------------------------------
// Tracking::SearchLocalPoints (src/Tracking.cc:1760-1830) - see shim/SearchLocalPoints.h.
// ---------------------------------------------------------------------------------------------
namespace
{
// mfMaxDistance / mfMinDistance are protected (include/MapPoint.h:231-232); GetMaxDistanceInvariance() returns 1.2f * mfMaxDistance, which
// does not divide back exactly.  Read under the mutex the getters take.
struct MapPointDistances : public MapPoint {
    static void Get(MapPoint *p, float &mx, float &mn)
    {
        MapPointDistances *q = static_cast<MapPointDistances *>(p);
        unique_lock<mutex> lock(q->mMutexPos);
        mx = q->mfMaxDistance; mn = q->mfMinDistance;
    }
};
}  // namespace

int SearchLocalPointsHIP(Frame &F, const std::vector<MapPoint *> &vpLocalMapPoints, int th, float nnratio, float viewingCosLimit)
{
    __atomic_add_fetch(&gLocalPointsCalls, 1, __ATOMIC_RELAXED);
    orbx_shim::Mark("SearchLocalPoints enters");
    for (std::vector<MapPoint *>::iterator vit = F.mvpMapPoints.begin(), vend = F.mvpMapPoints.end(); vit != vend; vit++) {     // step 1, :1765-1784
        MapPoint *pMP = *vit;
        if (!pMP) continue;
        if (MapPointAccess::BadElseIncreaseVisible(pMP)) *vit = static_cast<MapPoint *>(NULL);      // isBad() ? drop : IncreaseVisible(), one visit
        else { pMP->mnLastFrameSeen = F.mnId; pMP->mbTrackInView = false; }
    }
    // step 2 (:1791-1811): the points Frame::isInFrustum would be asked about, in list order, and what it and SearchByProjection read from them -
    // one visit per point under its two mutexes (shim/MapPointAccess.h) instead of seven getter calls and three cv::Mat clones
    const size_t L = vpLocalMapPoints.size();
    std::vector<int> idx;
    idx.reserve(L);
    std::vector<float> pos(L * 3 + 3), nrm(L * 3 + 3), mx(L + 1), mn(L + 1);
    std::vector<uint8_t> desc(L * 32 + 32), hasObs(L + 1);
    for (size_t i = 0; i < L; i++) {
        MapPoint *pMP = vpLocalMapPoints[i];
        if (pMP->mnLastFrameSeen == F.mnId) continue;
        const size_t k = idx.size();
        if (!MapPointAccess::Snapshot(pMP, &pos[3 * k], &nrm[3 * k], mx[k], mn[k], hasObs[k], &desc[32 * k])) continue;      // isBad()
        idx.push_back((int)i);
    }
    const int M = (int)idx.size(), N = F.N;
    if (M == 0) return 0;
    orbx_shim::Mark("SearchLocalPoints: map points marshalled");
    const int nFeat = N > 0 ? N : 1;
    std::vector<uint8_t> occupied((size_t)nFeat, 0), inView((size_t)M, 0);
    for (int i = 0; i < N; i++) occupied[(size_t)i] = (F.mvpMapPoints[(size_t)i] && F.mvpMapPoints[(size_t)i]->Observations() > 0) ? 1 : 0;   // src/ORBmatcher.cc:110-112
    std::vector<float> px((size_t)M), py((size_t)M), pxr((size_t)M), vc((size_t)M);
    std::vector<int32_t> lvl((size_t)M), assigned((size_t)nFeat, -1);
    float tcw[16], ratioTh[64];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) tcw[4 * r + c] = F.mTcw.at<float>(r, c);
    if (orbx_predict_scale_thresholds(F.mfLogScaleFactor, F.mnScaleLevels, ratioTh) != ORBX_OK)
        { orbx_shim::Fail("SearchLocalPoints"); return 0; }
    int32_t nmatches = 0;
    {
        orbx_projection_frame fr = {N > 0 ? (const orbx_keypoint *)&F.mvKeysUn[0] : 0, N > 0 ? F.mDescriptors.data : 0, N > 0 ? &F.mvuRight[0] : 0, &occupied[0], &N, nFeat, 1,
                                    Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
        orbx_frustum_frame pose = {tcw, Frame::fx, Frame::fy, Frame::cx, Frame::cy, F.mbf, Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY, ratioTh, F.mnScaleLevels, 1};
        orbx_local_points pts = {&pos[0], &nrm[0], &mx[0], &mn[0], &desc[0], &hasObs[0], M};
        orbx_shim::Mark("SearchLocalPoints: frame marshalled");
        if (orbx_search_local_points(Matcher(N > M ? N : M), &fr, &pose, &pts, &F.mvScaleFactors[0], (int)F.mvScaleFactors.size(), viewingCosLimit, (float)th, nnratio,
                                     &assigned[0], &nmatches, &inView[0], &px[0], &py[0], &pxr[0], &lvl[0], &vc[0]) != ORBX_OK)
            { orbx_shim::Fail("SearchLocalPoints"); return 0; }
    }
    orbx_shim::Mark("SearchLocalPoints: device call returned");
    for (int k = 0; k < M; k++) {                                                   // what Frame::isInFrustum leaves in the MapPoint, :615, :721-731
        MapPoint *pMP = vpLocalMapPoints[(size_t)idx[(size_t)k]];
        pMP->mbTrackInView = inView[(size_t)k] != 0;
        if (!inView[(size_t)k]) continue;
        pMP->mTrackProjX = px[(size_t)k]; pMP->mTrackProjXR = pxr[(size_t)k]; pMP->mTrackProjY = py[(size_t)k];
        pMP->mnTrackScaleLevel = lvl[(size_t)k]; pMP->mTrackViewCos = vc[(size_t)k];
        pMP->IncreaseVisible();                                                     // :1807
    }
    for (int i = 0; i < N; i++)
        if (assigned[(size_t)i] >= 0) F.mvpMapPoints[(size_t)i] = vpLocalMapPoints[(size_t)idx[(size_t)assigned[(size_t)i]]];     // src/ORBmatcher.cc:165
    orbx_shim::Mark("SearchLocalPoints returns");
    return nmatches;
}

}  // namespace ORB_SLAM2

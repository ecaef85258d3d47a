// shim/Optimizer_hip.cc -- HIP bodies for ORB_SLAM2::Optimizer::LocalBundleAdjustment and
// Optimizer::PoseOptimization.
//
//     void Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*)     src/Optimizer.cc:629-997
//     int  Optimizer::PoseOptimization(Frame*)                           src/Optimizer.cc:363-605
//     void Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust)   src/Optimizer.cc:86-360
//     void Optimizer::GlobalBundleAdjustemnt(pMap, ...)                  src/Optimizer.cc:55-84
// The window / correspondence collection and the write-back under the map mutex follow the reference
// line by line; the g2o block in between (graph construction, 5 + 10 Levenberg iterations with the
// Schur complement, outlier re-classification) is one orbx_lba_solve / orbx_pose_optimization call on
// flat arrays.  Precision at the boundary is float32 like the reference's cv::Mat (src/Converter.cc).
#include <algorithm>
#include <iostream>
#include <list>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "Optimizer.h"   // the reference's include/Optimizer.h; shim/Optimizer.h where g2o / Eigen are not installed
#include "orbx.h"
#include "shim_error.h"
#include "MapPointAccess.h"

static unsigned long gLbaCalls = 0, gPoseOptCalls = 0, gBaCalls = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_bundle_adjustment_calls(void) { return gBaCalls; }
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_lba_calls(void) { return gLbaCalls; }
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_pose_optimization_calls(void) { return gPoseOptCalls; }

namespace ORB_SLAM2
{

namespace
{
struct ThreadLba {
    orbx_lba *h;
    int kf, pt, ed;
    ThreadLba() : h(0), kf(0), pt(0), ed(0) {}
    ~ThreadLba() { if (h) orbx_lba_destroy(h); }
};
thread_local ThreadLba tLba;

struct ThreadPoseOpt {
    orbx_pose_optimizer *h;
    int cap;
    ThreadPoseOpt() : h(0), cap(0) {}
    ~ThreadPoseOpt() { if (h) orbx_pose_optimizer_destroy(h); }
};
thread_local ThreadPoseOpt tPose;

bool Fail(const char *what) { return orbx_shim::Fail((std::string("Optimizer::") + what).c_str()); }
// BundleAdjustment runs on a detached std::thread of LoopClosing (RunGlobalBundleAdjustment): an exception leaving it would end
// the process in std::terminate.  A failed call reports and leaves the map exactly as it was (what a g2o run that made no
// progress does as well).
bool Report(const char *what) { std::cerr << "Optimizer::" << what << " (orbx): " << orbx_last_error() << " - map left unchanged" << std::endl; return false; }

cv::Mat PoseMat(const float *p16)
{
    cv::Mat T(4, 4, CV_32F);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) T.at<float>(r, c) = p16[4 * r + c];
    return T;
}
}  // namespace

void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap)
{
    __atomic_add_fetch(&gLbaCalls, 1, __ATOMIC_RELAXED);
    // ---- local keyframes, local map points, fixed cameras: :634-694 verbatim in structure
    std::list<KeyFrame *> lLocalKeyFrames;
    lLocalKeyFrames.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    const std::vector<KeyFrame *> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
    for (size_t i = 0; i < vNeighKFs.size(); i++) {
        KeyFrame *pKFi = vNeighKFs[i];
        pKFi->mnBALocalForKF = pKF->mnId;
        if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
    }
    std::list<MapPoint *> lLocalMapPoints;
    for (std::list<KeyFrame *>::iterator lit = lLocalKeyFrames.begin(); lit != lLocalKeyFrames.end(); lit++) {
        std::vector<MapPoint *> vpMPs = (*lit)->GetMapPointMatches();
        for (std::vector<MapPoint *>::iterator vit = vpMPs.begin(); vit != vpMPs.end(); vit++) {
            MapPoint *pMP = *vit;
            if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) {
                lLocalMapPoints.push_back(pMP);
                pMP->mnBALocalForKF = pKF->mnId;
            }
        }
    }
    std::list<KeyFrame *> lFixedCameras;
    std::vector<std::pair<KeyFrame *, size_t> > observations;      // (GetObservations() copies a std::map per point: the observers in the same order, into one reused vector)
    for (std::list<MapPoint *>::iterator lit = lLocalMapPoints.begin(); lit != lLocalMapPoints.end(); lit++) {
        MapPointAccess::Observations(*lit, observations);
        for (std::vector<std::pair<KeyFrame *, size_t> >::iterator mit = observations.begin(); mit != observations.end(); mit++) {
            KeyFrame *pKFi = mit->first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
            }
        }
    }
    // ---- flat problem: keyframes = local then fixed (:708-738), edges in optimizer.addEdge order (:768-853)
    std::vector<KeyFrame *> kfs;
    std::map<KeyFrame *, int> kfIndex;
    std::vector<uint8_t> fixed;
    for (std::list<KeyFrame *>::iterator lit = lLocalKeyFrames.begin(); lit != lLocalKeyFrames.end(); lit++) {
        kfIndex[*lit] = (int)kfs.size(); kfs.push_back(*lit); fixed.push_back((*lit)->mnId == 0 ? 1 : 0);   // :722
    }
    for (std::list<KeyFrame *>::iterator lit = lFixedCameras.begin(); lit != lFixedCameras.end(); lit++) {
        kfIndex[*lit] = (int)kfs.size(); kfs.push_back(*lit); fixed.push_back(1);                             // :736
    }
    std::vector<float> poses(kfs.size() * 16), intr(kfs.size() * 5);
    for (size_t k = 0; k < kfs.size(); k++) {
        const cv::Mat Tcw = kfs[k]->GetPose();
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) poses[16 * k + 4 * r + c] = Tcw.at<float>(r, c);
        const float in5[5] = {kfs[k]->fx, kfs[k]->fy, kfs[k]->cx, kfs[k]->cy, kfs[k]->mbf};
        for (int i = 0; i < 5; i++) intr[5 * k + i] = in5[i];
    }
    std::vector<MapPoint *> mps(lLocalMapPoints.begin(), lLocalMapPoints.end());
    std::vector<float> points(mps.size() * 3), obs, invS2;
    std::vector<int32_t> ep, ek;
    std::vector<std::pair<KeyFrame *, MapPoint *> > edgeOwner;
    for (size_t l = 0; l < mps.size(); l++) {
        MapPointAccess::WorldPos(mps[l], &points[3 * l]);
        MapPointAccess::Observations(mps[l], observations);
        for (std::vector<std::pair<KeyFrame *, size_t> >::const_iterator mit = observations.begin(); mit != observations.end(); mit++) {
            KeyFrame *pKFi = mit->first;
            if (pKFi->isBad()) continue;                                                        // :783
            const std::map<KeyFrame *, int>::const_iterator kit = kfIndex.find(pKFi);
            if (kit == kfIndex.end()) continue;   // an observer that became visible after the window was collected (the reference would dereference a missing vertex)
            const cv::KeyPoint &kpUn = pKFi->mvKeysUn[mit->second];
            ep.push_back((int32_t)l); ek.push_back(kit->second);
            obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(pKFi->mvuRight[mit->second]);   // < 0: monocular edge (:788)
            invS2.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);
            edgeOwner.push_back(std::make_pair(pKFi, mps[l]));
        }
    }
    if (pbStopFlag && *pbStopFlag) return;                                                      // :858-860
    if (kfs.empty() || mps.empty() || ep.empty()) return;
    const int K = (int)kfs.size(), P = (int)mps.size(), E = (int)ep.size();
    if (!tLba.h || K > tLba.kf || P > tLba.pt || E > tLba.ed) {
        if (tLba.h) { orbx_lba_destroy(tLba.h); tLba.h = 0; }
        tLba.kf = std::max(2 * K, 64); tLba.pt = std::max(2 * P, 4096); tLba.ed = std::max(2 * E, 65536);
        if (orbx_lba_create(orbx_shim::Device(), tLba.kf, tLba.pt, tLba.ed, &tLba.h) != ORBX_OK) { tLba.h = 0; Fail("LocalBundleAdjustment"); return; }      // the map stays as it is
    }
    orbx_lba_problem prob = {K, &poses[0], &fixed[0], &intr[0], P, &points[0], E, &ep[0], &ek[0], &obs[0], &invS2[0]};
    std::vector<float> posesOut(poses.size()), pointsOut(points.size());
    std::vector<uint8_t> outlier((size_t)E);
    orbx_lba_result res = {&posesOut[0], &pointsOut[0], NULL, &outlier[0], {0}};
    if (orbx_lba_solve(tLba.h, &prob, (const volatile uint8_t *)pbStopFlag, &res) != ORBX_OK) { Fail("LocalBundleAdjustment"); return; }
    // ---- vToErase (:921-958) and write-back under the map mutex (:961-996)
    std::vector<std::pair<KeyFrame *, MapPoint *> > vToErase;
    for (int e = 0; e < E; e++)
        if (outlier[(size_t)e] && !edgeOwner[(size_t)e].second->isBad()) vToErase.push_back(edgeOwner[(size_t)e]);
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
    for (size_t i = 0; i < vToErase.size(); i++) {
        KeyFrame *pKFi = vToErase[i].first;
        MapPoint *pMPi = vToErase[i].second;
        pKFi->EraseMapPointMatch(pMPi);
        pMPi->EraseObservation(pKFi);
    }
    int k = 0;
    for (std::list<KeyFrame *>::iterator lit = lLocalKeyFrames.begin(); lit != lLocalKeyFrames.end(); lit++, k++)
        (*lit)->SetPose(PoseMat(&posesOut[16 * (size_t)k]));
    for (size_t l = 0; l < mps.size(); l++) {
        cv::Mat X(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) X.at<float>(i) = pointsOut[3 * l + i];
        mps[l]->SetWorldPos(X);
        mps[l]->UpdateNormalAndDepth();
    }
}

void Optimizer::GlobalBundleAdjustemnt(Map *pMap, int nIterations, bool *pbStopFlag, const unsigned long nLoopKF, const bool bRobust)
{
    std::vector<KeyFrame *> vpKFs = pMap->GetAllKeyFrames();       // :79-82
    std::vector<MapPoint *> vpMP = pMap->GetAllMapPoints();
    BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust);
}

void Optimizer::BundleAdjustment(const std::vector<KeyFrame *> &vpKFs, const std::vector<MapPoint *> &vpMP, int nIterations, bool *pbStopFlag,
                                 const unsigned long nLoopKF, const bool bRobust)
{
    __atomic_add_fetch(&gBaCalls, 1, __ATOMIC_RELAXED);
    std::vector<bool> vbNotIncludedMP(vpMP.size());
    // ---- vertices: every non-bad KeyFrame, keyframe 0 fixed (:120-132)
    std::vector<KeyFrame *> kfs;
    std::map<KeyFrame *, int> kfIndex;
    std::vector<uint8_t> fixed;
    long unsigned int maxKFid = 0;
    for (size_t i = 0; i < vpKFs.size(); i++) {
        KeyFrame *pKF = vpKFs[i];
        if (pKF->isBad()) continue;
        kfIndex[pKF] = (int)kfs.size(); kfs.push_back(pKF); fixed.push_back(pKF->mnId == 0 ? 1 : 0);
        if (pKF->mnId > maxKFid) maxKFid = pKF->mnId;
    }
    std::vector<float> poses(kfs.size() * 16), intr(kfs.size() * 5);
    for (size_t k = 0; k < kfs.size(); k++) {
        const cv::Mat Tcw = kfs[k]->GetPose();
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) poses[16 * k + 4 * r + c] = Tcw.at<float>(r, c);
        const float in5[5] = {kfs[k]->fx, kfs[k]->fy, kfs[k]->cx, kfs[k]->cy, kfs[k]->mbf};
        for (int i = 0; i < 5; i++) intr[5 * k + i] = in5[i];
    }
    // ---- points and edges in optimizer.addEdge order (:137-232); a point without edges is left out (:233-241)
    std::vector<MapPoint *> mps;
    std::vector<float> points, obs, invS2;
    std::vector<int32_t> ep, ek;
    for (size_t i = 0; i < vpMP.size(); i++) {
        MapPoint *pMP = vpMP[i];
        vbNotIncludedMP[i] = true;
        if (pMP->isBad()) continue;
        const std::map<KeyFrame *, size_t> observations = pMP->GetObservations();
        int nEdges = 0;
        const int l = (int)mps.size();
        for (std::map<KeyFrame *, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); mit++) {
            KeyFrame *pKF = mit->first;
            if (pKF->isBad() || pKF->mnId > maxKFid) continue;                                   // :156-157
            std::map<KeyFrame *, int>::const_iterator kit = kfIndex.find(pKF);
            if (kit == kfIndex.end()) continue;   // (the reference would dereference a missing vertex here; vpKFs always holds the observers)
            nEdges++;
            const cv::KeyPoint &kpUn = pKF->mvKeysUn[mit->second];
            ep.push_back((int32_t)l); ek.push_back(kit->second);
            obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(pKF->mvuRight[mit->second]);   // < 0: monocular edge (:164)
            invS2.push_back(pKF->mvInvLevelSigma2[kpUn.octave]);
        }
        if (nEdges == 0) continue;
        vbNotIncludedMP[i] = false;
        const cv::Mat X = pMP->GetWorldPos();
        for (int k = 0; k < 3; k++) points.push_back(X.at<float>(k));
        mps.push_back(pMP);
    }
    if (kfs.empty() || mps.empty() || ep.empty()) return;
    const int K = (int)kfs.size(), P = (int)mps.size(), E = (int)ep.size();
    if (!tLba.h || K > tLba.kf || P > tLba.pt || E > tLba.ed) {
        if (tLba.h) { orbx_lba_destroy(tLba.h); tLba.h = 0; }
        tLba.kf = std::max(2 * K, 64); tLba.pt = std::max(2 * P, 4096); tLba.ed = std::max(2 * E, 65536);
        if (orbx_lba_create(orbx_shim::Device(), tLba.kf, tLba.pt, tLba.ed, &tLba.h) != ORBX_OK) { Report("BundleAdjustment"); return; }
    }
    orbx_lba_problem prob = {K, &poses[0], &fixed[0], &intr[0], P, &points[0], E, &ep[0], &ek[0], &obs[0], &invS2[0]};
    std::vector<float> posesOut(poses.size()), pointsOut(points.size());
    std::vector<uint8_t> outlier((size_t)E);
    orbx_lba_result res = {&posesOut[0], &pointsOut[0], NULL, &outlier[0], {0}};
    if (orbx_bundle_adjustment(tLba.h, &prob, nIterations, bRobust ? 1 : 0, (const volatile uint8_t *)pbStopFlag, &res) != ORBX_OK) { Report("BundleAdjustment"); return; }
    // ---- write-back (:252-302); the bad flags are re-read here as the reference does (:258, :280): LocalMapping culls concurrently
    for (size_t k = 0; k < kfs.size(); k++) {
        KeyFrame *pKF = kfs[k];
        if (pKF->isBad()) continue;
        if (nLoopKF == 0) pKF->SetPose(PoseMat(&posesOut[16 * k]));
        else {
            pKF->mTcwGBA.create(4, 4, CV_32F);
            PoseMat(&posesOut[16 * k]).copyTo(pKF->mTcwGBA);
            pKF->mnBAGlobalForKF = nLoopKF;
        }
    }
    for (size_t l = 0; l < mps.size(); l++) {
        MapPoint *pMP = mps[l];
        if (pMP->isBad()) continue;
        cv::Mat X(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) X.at<float>(i) = pointsOut[3 * l + i];
        if (nLoopKF == 0) {
            pMP->SetWorldPos(X);
            pMP->UpdateNormalAndDepth();
        } else {
            pMP->mPosGBA.create(3, 1, CV_32F);
            X.copyTo(pMP->mPosGBA);
            pMP->mnBAGlobalForKF = nLoopKF;
        }
    }
}

int Optimizer::PoseOptimization(Frame *pFrame)
{
    __atomic_add_fetch(&gPoseOptCalls, 1, __ATOMIC_RELAXED);
    orbx_shim::Mark("PoseOptimization enters");
    const int N = pFrame->N;
    std::vector<int> index;   // features that have a MapPoint, in feature order (:396-500)
    std::vector<float> Xw, obs, invS2;
    index.reserve((size_t)N); Xw.reserve(3 * (size_t)N); obs.reserve(3 * (size_t)N); invS2.reserve((size_t)N);
    {
        std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);
        for (int i = 0; i < N; i++) {
            MapPoint *pMP = pFrame->mvpMapPoints[(size_t)i];
            if (!pMP) continue;
            pFrame->mvbOutlier[(size_t)i] = false;
            const cv::KeyPoint &kpUn = pFrame->mvKeysUn[(size_t)i];
            float X[3];
            MapPointAccess::WorldPos(pMP, X);      // GetWorldPos() without the cv::Mat clone
            index.push_back(i);
            for (int k = 0; k < 3; k++) Xw.push_back(X[k]);
            obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(pFrame->mvuRight[(size_t)i]);
            invS2.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
        }
    }
    const int n = (int)index.size();
    if (n < 3) return 0;                                                                        // :509-510
    if (!tPose.h || n > tPose.cap) {
        if (tPose.h) { orbx_pose_optimizer_destroy(tPose.h); tPose.h = 0; }
        tPose.cap = std::max(2 * n, 4096);
        if (orbx_pose_optimizer_create(orbx_shim::Device(), 1, tPose.cap, &tPose.h) != ORBX_OK) { tPose.h = 0; Fail("PoseOptimization"); return 0; }      // pose and outlier flags untouched, no inliers
    }
    float pose[16], cam[5] = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy, pFrame->mbf}, poseOut[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) pose[4 * r + c] = pFrame->mTcw.at<float>(r, c);
    orbx_pose_problem prob = {1, n, pose, cam, &n, &Xw[0], &obs[0], &invS2[0]};
    std::vector<uint8_t> outlier((size_t)n);
    int32_t inliers = 0;
    orbx_shim::Mark("PoseOptimization marshalled");
    if (orbx_pose_optimization(tPose.h, &prob, poseOut, &outlier[0], &inliers, NULL) != ORBX_OK) { Fail("PoseOptimization"); return 0; }
    orbx_shim::Mark("PoseOptimization device call returned");
    for (int e = 0; e < n; e++) pFrame->mvbOutlier[(size_t)index[(size_t)e]] = outlier[(size_t)e] != 0;
    pFrame->SetPose(PoseMat(poseOut));                                                          // :598-601
    orbx_shim::Mark("PoseOptimization returns");
    return inliers;
}

}  // namespace ORB_SLAM2

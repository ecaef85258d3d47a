// shim/MapPointAccess.h -- what the replaced bodies read from a MapPoint, in ONE visit per point.
//
// The reference's bodies reach a point's state through its getters: GetWorldPos / GetNormal / GetDescriptor clone a cv::Mat under a mutex each,
// isBad takes both mutexes, Observations / GetMin / MaxDistanceInvariance / IncreaseVisible one each (src/MapPoint.cc:77-81, 210-214, 310-331, 446-450,
// 523-533).  Frame::isInFrustum + SearchByProjection make seven such visits per local map point (src/Frame.cc:608-742, src/ORBmatcher.cc:70-175),
// and marshalling ~1900 points that way cost the drop-in 100 us of a 300 us SearchLocalPoints (profiles/r06_track_trace.txt).  The members are
// protected; a derived type reads them under the SAME two mutexes in the order MapPoint::isBad takes them (features, then position), copies nothing
// but the values, and never keeps a reference.  The reference's class is not changed.
#ifndef ORBX_SHIM_MAP_POINT_ACCESS_H
#define ORBX_SHIM_MAP_POINT_ACCESS_H

#include <map>
#include <mutex>
#include <string.h>
#include <utility>
#include <vector>

#include "MapPoint.h"

namespace ORB_SLAM2
{
struct MapPointAccess : public MapPoint {
    // isBad() and, for a point that is not: position, normal, distance range (the raw members: GetMaxDistanceInvariance() returns 1.2f * mfMaxDistance,
    // which does not divide back exactly), "has observations" and the representative descriptor.  false = bad (nothing written).
    static bool Snapshot(MapPoint *p, float pos[3], float nrm[3], float &maxD, float &minD, unsigned char &hasObs, unsigned char desc[32])
    {
        MapPointAccess *q = static_cast<MapPointAccess *>(p);
        std::unique_lock<std::mutex> lock(q->mMutexFeatures);
        std::unique_lock<std::mutex> lock2(q->mMutexPos);
        if (q->mbBad) return false;
        for (int c = 0; c < 3; c++) { pos[c] = q->mWorldPos.at<float>(c); nrm[c] = q->mNormalVector.at<float>(c); }      // 3x1 matrices
        maxD = q->mfMaxDistance; minD = q->mfMinDistance;
        hasObs = q->nObs > 0 ? 1 : 0;
        if (q->mDescriptor.data) memcpy(desc, q->mDescriptor.data, 32); else memset(desc, 0, 32);
        return true;
    }
    // GetWorldPos() without the clone
    static void WorldPos(MapPoint *p, float pos[3])
    {
        MapPointAccess *q = static_cast<MapPointAccess *>(p);
        std::unique_lock<std::mutex> lock(q->mMutexPos);
        for (int c = 0; c < 3; c++) pos[c] = q->mWorldPos.at<float>(c);
    }
    // step 1 of Tracking::SearchLocalPoints for a point a feature already holds (src/Tracking.cc:1765-1784): isBad(), else IncreaseVisible(); -> isBad
    static bool BadElseIncreaseVisible(MapPoint *p)
    {
        MapPointAccess *q = static_cast<MapPointAccess *>(p);
        std::unique_lock<std::mutex> lock(q->mMutexFeatures);
        std::unique_lock<std::mutex> lock2(q->mMutexPos);
        if (q->mbBad) return true;
        q->mnVisible += 1;
        return false;
    }
    // GetObservations() without the std::map copy (a node allocation per observer and point): the observers in the map's order, into a vector the caller reuses
    static void Observations(MapPoint *p, std::vector<std::pair<KeyFrame *, size_t> > &out)
    {
        MapPointAccess *q = static_cast<MapPointAccess *>(p);
        out.clear();
        std::unique_lock<std::mutex> lock(q->mMutexFeatures);
        for (std::map<KeyFrame *, size_t>::const_iterator it = q->mObservations.begin(); it != q->mObservations.end(); ++it) out.push_back(*it);
    }
    // isBad() and Observations() > 0 in one visit: 0 = bad, 1 = good without observations, 2 = good with
    static int GoodAndObserved(MapPoint *p)
    {
        MapPointAccess *q = static_cast<MapPointAccess *>(p);
        std::unique_lock<std::mutex> lock(q->mMutexFeatures);
        std::unique_lock<std::mutex> lock2(q->mMutexPos);
        return q->mbBad ? 0 : (q->nObs > 0 ? 2 : 1);
    }
};
}  // namespace ORB_SLAM2

#endif

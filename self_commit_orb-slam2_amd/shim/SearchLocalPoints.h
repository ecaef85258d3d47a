// shim/SearchLocalPoints.h -- device body for Tracking::SearchLocalPoints (src/Tracking.cc:1760-1830).
//
// Tracking.cc itself is outside this repository's scope; this is the binding a maintainer adds to it:
//
//     void Tracking::SearchLocalPoints()
//     {
//         int th = 1;                                                   // :1818-1825, unchanged
//         if (mSensor == System::RGBD) th = 3;
//         if (mCurrentFrame.mnId < mnLastRelocFrameId + 2) th = 5;
//         ORB_SLAM2::SearchLocalPointsHIP(mCurrentFrame, mvpLocalMapPoints, th);
//     }
//
// The function does what the reference's body does to the Frame and to the MapPoints - step 1 (features that already hold a MapPoint:
// IncreaseVisible, mnLastFrameSeen, mbTrackInView = false), Frame::isInFrustum for every other local point (mbTrackInView, mTrackProjX /
// Y / XR, mnTrackScaleLevel, mTrackViewCos; IncreaseVisible for the visible ones) and ORBmatcher(0.8).SearchByProjection(F, points, th) -
// with the per-point projection AND the search in one device call (orbx_search_local_points): the host only copies map data.
// Returns the number of new matches (the reference's function is void; its callers do not use a value).
#ifndef ORBX_SHIM_SEARCH_LOCAL_POINTS_H
#define ORBX_SHIM_SEARCH_LOCAL_POINTS_H

#include <vector>

#include "Frame.h"
#include "MapPoint.h"

namespace ORB_SLAM2
{
int SearchLocalPointsHIP(Frame &F, const std::vector<MapPoint *> &vpLocalMapPoints, int th, float nnratio = 0.8f, float viewingCosLimit = 0.5f);
}

#endif

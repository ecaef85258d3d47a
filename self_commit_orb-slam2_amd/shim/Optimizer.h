// shim/Optimizer.h -- the static entry points of ORB_SLAM2::Optimizer that liborbx implements.
//
// The reference's include/Optimizer.h (lines 36-66) pulls in g2o headers (Eigen).  A maintainer keeps
// that header and only swaps these function BODIES for shim/Optimizer_hip.cc; this minimal
// declaration exists so that the bodies can be compiled and tested where g2o / Eigen are not
// installed (the drop-in test build defines OPTIMIZER_H so the reference header is skipped).
#ifndef ORBX_SHIM_OPTIMIZER_H
#define ORBX_SHIM_OPTIMIZER_H

#include "Frame.h"
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"

namespace ORB_SLAM2
{

class Optimizer
{
public:
    // reference include/Optimizer.h:112, src/Optimizer.cc:629-997
    void static LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap);
    // reference include/Optimizer.h:94, src/Optimizer.cc:363-605
    int static PoseOptimization(Frame *pFrame);
    // reference include/Optimizer.h:59-66 / 76-79, src/Optimizer.cc:86-360 / 55-84
    void static BundleAdjustment(const std::vector<KeyFrame *> &vpKF, const std::vector<MapPoint *> &vpMP, int nIterations = 5, bool *pbStopFlag = NULL,
                                 const unsigned long nLoopKF = 0, const bool bRobust = true);
    void static GlobalBundleAdjustemnt(Map *pMap, int nIterations = 5, bool *pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
};

}  // namespace ORB_SLAM2

#endif

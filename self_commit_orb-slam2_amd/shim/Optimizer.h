// shim/Optimizer.h -- the two static entry points of ORB_SLAM2::Optimizer that liborbx implements.
//
// The reference's include/Optimizer.h (lines 36-66) pulls in g2o headers (Eigen).  A maintainer keeps
// that header and only swaps the two function BODIES for shim/Optimizer_hip.cc; this minimal
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
};

}  // namespace ORB_SLAM2

#endif

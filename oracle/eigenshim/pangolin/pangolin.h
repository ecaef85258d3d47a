// oracle/eigenshim/pangolin/pangolin.h -- TEST INFRASTRUCTURE ONLY.
// include/Optimizer.h pulls in LoopClosing.h -> Tracking.h -> Viewer.h / MapDrawer.h, which name
// pangolin::OpenGlMatrix in two declarations (include/MapDrawer.h:73,91).  Nothing of the viewer
// is compiled; this is the one type those declarations need.
#pragma once
namespace pangolin {
struct OpenGlMatrix { double m[16]; };
}

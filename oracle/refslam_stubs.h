// oracle/refslam_stubs.h -- TEST INFRASTRUCTURE ONLY.
// Force-included (-include) when the reference sources are compiled into oracle/_ref/liborbslam.so
// with -DCONVERTER_H: the reference's include/Converter.h pulls in Eigen and the g2o types, which
// do not exist in this container; the compiled files (src/Frame.cc:884, src/KeyFrame.cc:84) only
// call Converter::toDescriptorVector, declared here and defined in oracle/refslam_wrap.cc.
#pragma once
#include <vector>
#include <opencv2/core/core.hpp>
namespace ORB_SLAM2 {
class Converter {
public:
    static std::vector<cv::Mat> toDescriptorVector(const cv::Mat &Descriptors);
};
}

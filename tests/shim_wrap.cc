// tests/shim_wrap.cc -- C wrapper around the drop-in ORB_SLAM2::ORBextractor of shim/,
// built against oracle/cvshim (this box has no OpenCV) + liborbx.so.  The calling code is the
// reference's own call shape, Frame::ExtractORB (src/Frame.cc:503): (*extractor)(im, cv::Mat(), keys, desc).
#include <cstring>
#include <vector>

#include "ORBextractor.h"

extern "C" void *shim_create(int nf, float sf, int nl, int ini, int mn)
{
    try { return new ORB_SLAM2::ORBextractor(nf, sf, nl, ini, mn); } catch (...) { return 0; }
}
extern "C" void shim_destroy(void *h) { delete (ORB_SLAM2::ORBextractor *)h; }

extern "C" int shim_extract(void *h, const unsigned char *img, int w, int hgt, int stride, float *kps, unsigned char *desc, int cap)
{
    ORB_SLAM2::ORBextractor *e = (ORB_SLAM2::ORBextractor *)h;
    cv::Mat im(hgt, w, CV_8UC1, (void *)img, (size_t)stride);
    std::vector<cv::KeyPoint> keys;
    cv::Mat d;
    (*e)(im, cv::Mat(), keys, d);
    int n = (int)keys.size();
    for (int i = 0; i < n && i < cap; i++) {
        float *o = kps + 7 * (size_t)i;
        o[0] = keys[i].pt.x; o[1] = keys[i].pt.y; o[2] = keys[i].size; o[3] = keys[i].angle; o[4] = keys[i].response;
        o[5] = (float)keys[i].octave; o[6] = (float)keys[i].class_id;
        memcpy(desc + 32 * (size_t)i, d.ptr(i), 32);
    }
    return n;
}

extern "C" int shim_pyramid_level(void *h, int level, unsigned char *dst, int *w, int *hgt)
{
    ORB_SLAM2::ORBextractor *e = (ORB_SLAM2::ORBextractor *)h;
    const cv::Mat &m = e->mvImagePyramid[level];
    *w = m.cols; *hgt = m.rows;
    for (int y = 0; y < m.rows; y++) memcpy(dst + (size_t)y * m.cols, m.ptr(y), (size_t)m.cols);
    return 0;
}

extern "C" int shim_levels(void *h) { return ((ORB_SLAM2::ORBextractor *)h)->GetLevels(); }

// tests/shim_wrap.cc -- C wrapper around the drop-in ORB_SLAM2::ORBextractor of shim/,
// built against oracle/cvshim (this box has no OpenCV) + liborbx.so.  The calling code is the
// reference's own call shape, Frame::ExtractORB (src/Frame.cc:503): (*extractor)(im, cv::Mat(), keys, desc).
#include <cstring>
#include <stdexcept>
#include <vector>

#include "ORBextractor.h"
#include "orbx.h"
#include <cstdio>
#include <cstdlib>

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
    if (level == 0) e->DownloadImagePyramid();      // on demand (the copy is off by default)
    const cv::Mat &m = e->mvImagePyramid[level];
    *w = m.cols; *hgt = m.rows;
    for (int y = 0; y < m.rows; y++) memcpy(dst + (size_t)y * m.cols, m.ptr(y), (size_t)m.cols);
    return 0;
}

// The member binds like the reference's `std::vector<cv::Mat> mvImagePyramid` (include/ORBextractor.h:161): a reference to the vector, iteration, at(),
// front() / back(), size() / empty() - each of them sees the CURRENT frame's levels.  Returns a checksum over all the routes (they must agree), < 0 on a mismatch.
static long level_sum(const cv::Mat &m) { long s = m.rows * 131 + m.cols; for (int y = 0; y < m.rows; y += 7) s += m.ptr(y)[(y * 3) % (m.cols > 0 ? m.cols : 1)]; return s; }
extern "C" long shim_pyramid_binds_like_a_vector(void *h)
{
    ORB_SLAM2::ORBextractor *e = (ORB_SLAM2::ORBextractor *)h;
    std::vector<cv::Mat> &v = e->mvImagePyramid;                      // what a caller written against the reference does
    const std::vector<cv::Mat> &cv_ = e->mvImagePyramid;
    long a = 0, b = 0, c = 0, d = 0;
    for (size_t l = 0; l < v.size(); l++) a += level_sum(v[l]);
    for (auto &m : e->mvImagePyramid) b += level_sum(m);
    for (ORB_SLAM2::ORBextractor::ImagePyramid::const_iterator it = e->mvImagePyramid.begin(); it != e->mvImagePyramid.end(); ++it) c += level_sum(*it);
    for (size_t l = 0; l < e->mvImagePyramid.size(); l++) d += level_sum(e->mvImagePyramid.at(l));
    if (a != b || a != c || a != d || cv_.size() != v.size() || e->mvImagePyramid.empty()) return -1;
    if (level_sum(e->mvImagePyramid.front()) != level_sum(v[0]) || level_sum(e->mvImagePyramid.back()) != level_sum(v[v.size() - 1])) return -2;
    try { (void)e->mvImagePyramid.at(v.size()); return -3; } catch (const std::out_of_range &) {}
    return a;
}
// ... and none of that needs a device to COMPILE and link (the CPU test): an extractor that never ran has nlevels empty levels
extern "C" int shim_pyramid_of_a_fresh_extractor(void *h)
{
    ORB_SLAM2::ORBextractor *e = (ORB_SLAM2::ORBextractor *)h;
    std::vector<cv::Mat> &v = e->mvImagePyramid;
    int n = 0;
    for (auto &m : e->mvImagePyramid) n += m.empty() ? 1 : 100;
    e->mvImagePyramid.clear();
    return (int)v.size() * 1000 + n;
}

extern "C" int shim_levels(void *h) { return ((ORB_SLAM2::ORBextractor *)h)->GetLevels(); }

// what the reference's own Frame::ComputeStereoMatches would find in the public member right after operator() (src/Frame.cc:1044,1248): the
// flag's default, and the size of a level WITHOUT an explicit DownloadImagePyramid()
extern "C" int shim_keeps_host_pyramid(void *h) { return ((ORB_SLAM2::ORBextractor *)h)->mbKeepHostPyramid ? 1 : 0; }
extern "C" int shim_pyramid_level_as_is(void *h, int level, unsigned char *dst, int *w, int *hgt)
{
    const cv::Mat &m = ((ORB_SLAM2::ORBextractor *)h)->mvImagePyramid[level];
    *w = m.cols; *hgt = m.rows;
    for (int y = 0; y < m.rows; y++) memcpy(dst + (size_t)y * m.cols, m.ptr(y), (size_t)m.cols);
    return 0;
}

// A level somebody kept from an earlier frame (a cv::Mat header copy, as a caller of the reference may hold one: its levels are fresh Mats per
// call, src/ORBextractor.cc:1687-1689) must still hold THAT frame after later calls, after the handle was rebuilt for a larger image, and after
// the extractor is gone.  Returns the number of bytes of `expect` (level `level` of the first image, w x hgt) that the kept level no longer has.
extern "C" long shim_kept_level_survives(int nf, const unsigned char *img1, const unsigned char *img2, int w, int hgt, const unsigned char *big, int bw, int bh, int level,
                                         const unsigned char *expect, int ew, int eh)
{
    ORB_SLAM2::ORBextractor *e = new ORB_SLAM2::ORBextractor(nf, 1.2f, 8, 20, 7);
    e->mbKeepHostPyramid = true;
    std::vector<cv::KeyPoint> keys;
    cv::Mat d;
    cv::Mat a(hgt, w, CV_8UC1, (void *)img1, (size_t)w), b(hgt, w, CV_8UC1, (void *)img2, (size_t)w), c(bh, bw, CV_8UC1, (void *)big, (size_t)bw);
    (*e)(a, cv::Mat(), keys, d);
    cv::Mat kept = e->mvImagePyramid[(size_t)level];      // header copy: shares the level's buffer
    (*e)(b, cv::Mat(), keys, d);
    (void)e->mvImagePyramid[(size_t)level].rows;          // the next frame's pyramid is read too (a fresh set of levels)
    (*e)(c, cv::Mat(), keys, d);                          // larger image: the liborbx handle is destroyed and rebuilt
    (void)e->mvImagePyramid[0].rows;
    delete e;
    if (kept.cols != ew || kept.rows != eh) return -1;
    long bad = 0;
    for (int y = 0; y < eh; y++)
        for (int x = 0; x < ew; x++) bad += kept.ptr(y)[x] != expect[(size_t)y * ew + x];
    return bad;
}

// Latency of the reference's call shape, one frame per call on one thread (tools/latency_shim.py): mean / median microseconds of
// `iters` calls of (*extractor)(im, cv::Mat(), keys, desc) cycling through `nimg` images; keep_pyr sets mbKeepHostPyramid.
#include <algorithm>
#include <chrono>
extern "C" int shim_bench(void *h, const unsigned char *const *imgs, int nimg, int w, int hgt, int stride, int iters, int keep_pyr, double *mean_us,
                          double *median_us, int *nkeys)
{
    ORB_SLAM2::ORBextractor *e = (ORB_SLAM2::ORBextractor *)h;
    e->mbKeepHostPyramid = keep_pyr != 0;
    e->mbViewHostPyramid = keep_pyr == 3;      // 3: as 1 (read after every call) with the opt-in views instead of owning copies
    if (keep_pyr == 3) keep_pyr = 1;
    std::vector<cv::KeyPoint> keys;
    cv::Mat d;
    std::vector<double> t((size_t)iters);
    for (int i = -5; i < iters; i++) {
        cv::Mat im(hgt, w, CV_8UC1, (void *)imgs[(i + 5) % nimg], (size_t)stride);
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        (*e)(im, cv::Mat(), keys, d);
        if (keep_pyr == 1) {      // the pyramid is READ after every call, as the reference's stereo ComputeStereoMatches does (src/Frame.cc:1044, 1248): inside the timed span
            unsigned acc = 0;
            for (int l = 0; l < e->GetLevels(); l++) { const cv::Mat &m = e->mvImagePyramid[(size_t)l]; acc += (unsigned)m.rows + m.ptr(m.rows - 1)[m.cols - 1]; }
            if (acc == 0xffffffffu) keys.clear();
        }
        const std::chrono::steady_clock::time_point t1 = std::chrono::steady_clock::now();
        if (i >= 0) t[(size_t)i] = std::chrono::duration<double, std::micro>(t1 - t0).count();
    }
    double s = 0;
    for (int i = 0; i < iters; i++) s += t[(size_t)i];
    std::sort(t.begin(), t.end());
    *mean_us = s / iters; *median_us = t[(size_t)iters / 2]; *nkeys = (int)keys.size();
    return 0;
}

// The multi-handle recipe: `nthreads` threads, each with its OWN ORBextractor (instances are not re-entrant - the reference's are not
// either, include/ORBextractor.h:161: the stereo Frame constructor runs two on two threads, src/Frame.cc:159-167), each calling
// operator() `iters` times back to back.  Returns the aggregate frames/s.
#include <thread>
extern "C" double shim_bench_threads(int nthreads, int nf, const unsigned char *const *imgs, int nimg, int w, int hgt, int stride, int iters, int keep_pyr)
{
    std::vector<ORB_SLAM2::ORBextractor *> ex;
    for (int t = 0; t < nthreads; t++) { ex.push_back(new ORB_SLAM2::ORBextractor(nf, 1.2f, 8, 20, 7)); ex.back()->mbKeepHostPyramid = keep_pyr != 0; ex.back()->mbViewHostPyramid = keep_pyr == 3; }
    if (keep_pyr == 3) keep_pyr = 1;
    auto work = [&](int t, int n) {
        std::vector<cv::KeyPoint> keys;
        cv::Mat d;
        for (int i = 0; i < n; i++) {
            cv::Mat im(hgt, w, CV_8UC1, (void *)imgs[(i + t) % nimg], (size_t)stride);
            (*ex[(size_t)t])(im, cv::Mat(), keys, d);
            if (keep_pyr == 1) {      // read every level after every call (see shim_bench); 2 = kept but never read
                unsigned acc = 0;
                ORB_SLAM2::ORBextractor *e = ex[(size_t)t];
                for (int l = 0; l < e->GetLevels(); l++) { const cv::Mat &m = e->mvImagePyramid[(size_t)l]; acc += (unsigned)m.rows + m.ptr(m.rows - 1)[m.cols - 1]; }
                if (acc == 0xffffffffu) keys.clear();
            }
        }
    };
    { std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(work, t, 40); for (auto &x : th) x.join(); }   // warm-up (the launch-set graphs of the sizes this load produces are built on first use)
    orbx_combiner_reset_stats(ex[0]->Handle());
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    { std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(work, t, iters); for (auto &x : th) x.join(); }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("ORBX_SHIM_BENCH_STATS")) {      // how the calls met inside liborbx: launch sets, frames, engines
        long long nb = 0, nfr = 0;
        int ne = 0;
        orbx_combiner_stats(ex[0]->Handle(), (int64_t *)&nb, (int64_t *)&nfr, &ne);
        double us[4] = {0, 0, 0, 0};
        orbx_combiner_profile(ex[0]->Handle(), us);
        fprintf(stderr, "[shim_bench_threads] %d threads: %lld frames in %lld launch sets (mean %.2f), %d engine(s); since the engines exist: staging %.1f us/frame, "
                "leader wait %.1f, launch call %.1f, device + sync %.1f us/set; wall %.1f us/frame\n", nthreads, nfr, nb, nb ? (double)nfr / nb : 0.0, ne,
                nfr ? us[0] / nfr : 0.0, nb ? us[1] / nb : 0.0, nb ? us[2] / nb : 0.0, nb ? us[3] / nb : 0.0, 1e6 * s / ((double)nthreads * iters));
        int64_t sets[33];
        double mus[33];
        orbx_combiner_histogram(ex[0]->Handle(), 32, sets, mus);
        fprintf(stderr, "    sets by size (n: count @ mean us):");
        for (int n = 1; n <= 32; n++) if (sets[n]) fprintf(stderr, " %d: %lld @ %.0f;", n, (long long)sets[n], mus[n]);
        fprintf(stderr, "\n");
    }
    for (size_t t = 0; t < ex.size(); t++) delete ex[t];
    return (double)nthreads * iters / s;
}

// Error channel of the drop-in class (shim/ORBextractor.h: ErrorCount / LastError / Dead / sbThrowOnError).  `stale` keypoints are put into
// the caller's vector before the call: a failed call must hand back EMPTY outputs, never the previous frame's.
extern "C" int shim_error_count(void *h) { return ((ORB_SLAM2::ORBextractor *)h)->ErrorCount(); }
extern "C" int shim_dead(void *h) { return ((ORB_SLAM2::ORBextractor *)h)->Dead() ? 1 : 0; }
extern "C" const char *shim_last_error(void *h) { return ((ORB_SLAM2::ORBextractor *)h)->LastError().c_str(); }
extern "C" void shim_set_throw(int on) { ORB_SLAM2::ORBextractor::sbThrowOnError = on != 0; }
extern "C" int shim_extract_over_stale_outputs(void *h, const unsigned char *img, int w, int hgt, int stride, int stale, int *desc_rows)
{
    ORB_SLAM2::ORBextractor *e = (ORB_SLAM2::ORBextractor *)h;
    cv::Mat im(hgt, w, CV_8UC1, (void *)img, (size_t)stride);
    std::vector<cv::KeyPoint> keys((size_t)stale);
    cv::Mat d(stale > 0 ? stale : 1, 32, CV_8U);
    int thrown = 0;
    try { (*e)(im, cv::Mat(), keys, d); } catch (const std::exception &) { thrown = 1; }
    *desc_rows = d.rows;
    return thrown ? -1 - (int)keys.size() : (int)keys.size();
}

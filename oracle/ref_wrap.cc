// oracle/ref_wrap.cc -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// C wrapper around the UNMODIFIED reference class ORB_SLAM2::ORBextractor
// (/root/reference/src/ORBextractor.cc, include/ORBextractor.h) compiled against
// oracle/cvshim.  Built by oracle/Makefile into oracle/_ref/liborbref.so; used by
// tests/ and by bench.py's cpu_baseline leg, never by the product path.
//
// Determinism: DistributeOctTree sorts pair<int,ExtractorNode*> (reference
// src/ORBextractor.cc:948), i.e. equal point counts are ordered by heap address.
// This library replaces operator new with a per-thread monotone bump arena while a
// frame is being processed, so addresses increase with allocation order and the
// untouched reference code realises the project's tie rule "among equal counts the
// later-created node is split first" (SURVEY.md Appendix B).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ORBextractor.h"

namespace {
struct Arena {
    char *base;
    size_t cap, off;
    bool active;
};
thread_local Arena g_arena = {nullptr, 0, 0, false};
const size_t kArenaBytes = (size_t)1 << 31;   // 2 GiB of address space per thread, touched lazily

// orbref_set_arena(0): leave the allocator alone (stock glibc malloc, what a real ORB-SLAM2 binary runs with) - used only to
// MEASURE how far the allocator-dependent tie-break of src/ORBextractor.cc:948 moves the result (tests/test_tie_rule.py).
bool g_use_arena = true;

void arena_begin()
{
    if (!g_use_arena) { g_arena.active = false; return; }
    if (!g_arena.base) {
        g_arena.base = (char *)malloc(kArenaBytes);
        if (!g_arena.base) { fprintf(stderr, "orbref: arena malloc failed\n"); abort(); }
        g_arena.cap = kArenaBytes;
    }
    g_arena.off = 0;
    g_arena.active = true;
}
void arena_end() { g_arena.active = false; }
inline bool in_arena(void *p) { return g_arena.base && (char *)p >= g_arena.base && (char *)p < g_arena.base + g_arena.cap; }
}  // namespace

#define ORBREF_HIDDEN __attribute__((visibility("hidden")))
ORBREF_HIDDEN void *operator new(size_t n)
{
    if (g_arena.active) {
        size_t a = (g_arena.off + 15) & ~(size_t)15;
        if (a + n > g_arena.cap) { fprintf(stderr, "orbref: arena exhausted\n"); abort(); }
        g_arena.off = a + n;
        return g_arena.base + a;
    }
    void *p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
ORBREF_HIDDEN void *operator new[](size_t n) { return operator new(n); }
ORBREF_HIDDEN void operator delete(void *p) noexcept { if (p && !in_arena(p)) free(p); }
ORBREF_HIDDEN void operator delete[](void *p) noexcept { operator delete(p); }
ORBREF_HIDDEN void operator delete(void *p, size_t) noexcept { operator delete(p); }
ORBREF_HIDDEN void operator delete[](void *p, size_t) noexcept { operator delete(p); }

namespace {
// exposes the protected stages of the reference class
class RefExtractor : public ORB_SLAM2::ORBextractor {
public:
    RefExtractor(int nf, float sf, int nl, int ini, int mn) : ORB_SLAM2::ORBextractor(nf, sf, nl, ini, mn) {}
    void pyramid(const cv::Mat &im) { ComputePyramid(im); }
    void keypoints(std::vector<std::vector<cv::KeyPoint> > &all) { ComputeKeyPointsOctTree(all); }
    std::vector<cv::KeyPoint> octree(const std::vector<cv::KeyPoint> &c, int minX, int maxX, int minY, int maxY, int N)
    {
        return DistributeOctTree(c, minX, maxX, minY, maxY, N, 0);
    }
    const std::vector<int> &quotas() const { return mnFeaturesPerLevel; }
    const std::vector<int> &umax_() const { return umax; }
};
}  // namespace

#define ORBREF_API extern "C" __attribute__((visibility("default")))

ORBREF_API void orbref_set_arena(int on) { g_use_arena = on != 0; }

ORBREF_API void *orbref_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
{
    return new RefExtractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
}
ORBREF_API void orbref_destroy(void *h) { delete (RefExtractor *)h; }

// tables[0..nl) scale, [nl..2nl) invScale, [2nl..3nl) sigma2, [3nl..4nl) invSigma2; quotas[nl]; umax[16]
ORBREF_API void orbref_tables(void *h, float *tables, int *quotas, int *umax)
{
    RefExtractor *e = (RefExtractor *)h;
    int nl = e->GetLevels();
    std::vector<float> a = e->GetScaleFactors(), b = e->GetInverseScaleFactors(), c = e->GetScaleSigmaSquares(),
                       d = e->GetInverseScaleSigmaSquares();
    for (int i = 0; i < nl; i++) {
        tables[i] = a[i]; tables[nl + i] = b[i]; tables[2 * nl + i] = c[i]; tables[3 * nl + i] = d[i];
        quotas[i] = e->quotas()[i];
    }
    for (int i = 0; i < 16; i++) umax[i] = e->umax_()[i];
}

// Full operator(): keypoints as 7 floats each {x,y,size,angle,response,octave,class_id}.
// Returns the real keypoint count (may exceed cap; only cap entries are written).
ORBREF_API int orbref_extract(void *h, const uint8_t *img, int w, int hgt, int stride, float *kps, uint8_t *desc, int cap)
{
    RefExtractor *e = (RefExtractor *)h;
    int n = 0;
    arena_begin();
    {
        cv::Mat im(hgt, w, CV_8UC1, (void *)img, (size_t)stride);
        std::vector<cv::KeyPoint> keys;
        cv::Mat d;
        (*e)(im, cv::Mat(), keys, d);
        n = (int)keys.size();
        for (int i = 0; i < n && i < cap; i++) {
            const cv::KeyPoint &k = keys[i];
            float *o = kps + 7 * (size_t)i;
            o[0] = k.pt.x; o[1] = k.pt.y; o[2] = k.size; o[3] = k.angle; o[4] = k.response;
            o[5] = (float)k.octave; o[6] = (float)k.class_id;
            memcpy(desc + 32 * (size_t)i, d.ptr(i), 32);
        }
    }
    arena_end();
    return n;
}

// Stage access: pyramid + per-level keypoints (level coordinates, before descriptors).
ORBREF_API int orbref_keypoints(void *h, const uint8_t *img, int w, int hgt, int stride, float *kps, int *level_counts, int cap)
{
    RefExtractor *e = (RefExtractor *)h;
    int n = 0;
    arena_begin();
    {
        cv::Mat im(hgt, w, CV_8UC1, (void *)img, (size_t)stride);
        e->pyramid(im);
        std::vector<std::vector<cv::KeyPoint> > all;
        e->keypoints(all);
        for (size_t l = 0; l < all.size(); l++) {
            level_counts[l] = (int)all[l].size();
            for (size_t i = 0; i < all[l].size(); i++, n++) {
                if (n >= cap) continue;
                const cv::KeyPoint &k = all[l][i];
                float *o = kps + 7 * (size_t)n;
                o[0] = k.pt.x; o[1] = k.pt.y; o[2] = k.size; o[3] = k.angle; o[4] = k.response;
                o[5] = (float)k.octave; o[6] = (float)k.class_id;
            }
        }
    }
    arena_end();
    return n;
}

// Copy out pyramid level `level` of the last processed frame (tight rows).
ORBREF_API int orbref_pyramid_level(void *h, int level, uint8_t *dst, int *w, int *hgt)
{
    RefExtractor *e = (RefExtractor *)h;
    if (level < 0 || level >= (int)e->mvImagePyramid.size() || e->mvImagePyramid[level].empty()) return -1;
    const cv::Mat &m = e->mvImagePyramid[level];
    *w = m.cols; *hgt = m.rows;
    if (dst) for (int y = 0; y < m.rows; y++) memcpy(dst + (size_t)y * m.cols, m.ptr(y), (size_t)m.cols);
    return 0;
}

// DistributeOctTree alone: candidates {x,y,response} floats relative to (minX,minY).
ORBREF_API int orbref_octree(void *h, const float *cand, int ncand, int minX, int maxX, int minY, int maxY, int N, float *out, int cap)
{
    RefExtractor *e = (RefExtractor *)h;
    int n = 0;
    arena_begin();
    {
        std::vector<cv::KeyPoint> c((size_t)ncand);
        for (int i = 0; i < ncand; i++) c[i] = cv::KeyPoint(cand[3 * i], cand[3 * i + 1], 7.f, -1, cand[3 * i + 2]);
        std::vector<cv::KeyPoint> r = e->octree(c, minX, maxX, minY, maxY, N);
        n = (int)r.size();
        for (int i = 0; i < n && i < cap; i++) { out[3 * i] = r[i].pt.x; out[3 * i + 1] = r[i].pt.y; out[3 * i + 2] = r[i].response; }
    }
    arena_end();
    return n;
}

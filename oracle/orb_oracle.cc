// oracle/orb_oracle.cc -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Independent CPU restatement of ORB_SLAM2::ORBextractor, stage by stage, on flat
// arrays.  It is NOT the product: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load liborb_oracle.so.  Each function cites the reference code
// it follows (paths relative to /root/reference).
//
// How it is pinned: the reference ships no tests or golden
// vectors for this path, so this restatement is checked against oracle/_ref/
// liborbref.so = the UNMODIFIED reference src/ORBextractor.cc compiled against
// oracle/cvshim (tests/test_oracle_vs_ref.py: pyramid bytes, per-level keypoints,
// angles as bit patterns, descriptors, full operator() output).  The OpenCV
// primitives underneath both are oracle/prims.h ("parity unpinned" vs a real OpenCV,
// which is not available in this environment).
//
// The quadtree is restated in the array form the HIP kernel uses (list = array,
// children partition the parent's point span in place) rather than with std::list;
// equality with the std::list original is what test_oracle_vs_ref checks.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "prims.h"

namespace {

const int kPatchSize = 31;       // src/ORBextractor.cc:91
const int kHalfPatch = 15;       // :92
const int kEdgeThreshold = 19;   // :93
const float kCellW = 30.f;       // :1060

const int8_t kPattern[1024] = {
#include "orb_pattern.inc"
};

struct Oracle {
    int nfeatures, nlevels, iniTh, minTh;
    double scaleFactor;   // member is double initialised from a float (include/ORBextractor.h:206)
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> quota;
    int umax[16];
};

// ORBextractor::ORBextractor, src/ORBextractor.cc:492-609
void build_tables(Oracle &o)
{
    int nl = o.nlevels;
    o.scale.resize(nl); o.sigma2.resize(nl); o.invScale.resize(nl); o.invSigma2.resize(nl); o.quota.resize(nl);
    o.scale[0] = 1.0f; o.sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        o.scale[i] = (float)(o.scale[i - 1] * o.scaleFactor);   // float*double -> float (:512)
        o.sigma2[i] = o.scale[i] * o.scale[i];
    }
    for (int i = 0; i < nl; i++) { o.invScale[i] = 1.0f / o.scale[i]; o.invSigma2[i] = 1.0f / o.sigma2[i]; }
    float factor = (float)(1.0f / o.scaleFactor);                                                   // :538
    float nDesired = o.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));   // :539
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        o.quota[l] = op_round_f(nDesired);
        sum += o.quota[l];
        nDesired *= factor;
    }
    o.quota[nl - 1] = std::max(o.nfeatures - sum, 0);
    // umax, :579-608
    int v, v0, vmax = op_floor_d(kHalfPatch * sqrt(2.f) / 2 + 1), vmin = op_ceil_d(kHalfPatch * sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= vmax; ++v) o.umax[v] = op_round_d(sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (o.umax[v0] == o.umax[v0 + 1]) ++v0;
        o.umax[v] = v0;
        ++v0;
    }
}

// ComputePyramid level sizes, src/ORBextractor.cc:1680-1682
void level_size(const Oracle &o, int W, int H, int level, int *w, int *h)
{
    float s = o.invScale[level];
    *w = op_round_f((float)W * s);
    *h = op_round_f((float)H * s);
}

// One FAST score byte per pixel at threshold minTh (0 = not a corner at minTh).  A corner
// at iniTh is exactly a pixel with score >= iniTh (cornerScore<16> is threshold
// independent for true corners), so this single map serves both cv::FAST calls of
// src/ORBextractor.cc:1126,1135.  Arc-min definition of the score (independent of
// prims.h's early-out form, which the cvshim build uses).
uint8_t score_at(const uint8_t *p, size_t step, int minTh)
{
    int v = p[0], d[25];
    for (int k = 0; k < 25; k++) {
        int kk = k & 15;
        d[k] = v - p[(ptrdiff_t)OP_FAST_DY[kk] * (ptrdiff_t)step + OP_FAST_DX[kk]];
    }
    int best = -1000;   // max over the 16 arcs of 9 of min(+d) and min(-d)
    for (int s = 0; s < 16; s++) {
        int mn = 1000, mx = -1000;
        for (int k = s; k < s + 9; k++) { mn = std::min(mn, d[k]); mx = std::max(mx, d[k]); }
        best = std::max(best, std::max(mn, -mx));
    }
    int score = best - 1;   // largest t with all 9 strictly beyond t
    return (uint8_t)(score >= minTh ? score : 0);
}

void score_map(const uint8_t *img, int w, int h, size_t step, int minTh, uint8_t *out /* w*h tight */)
{
    memset(out, 0, (size_t)w * h);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) out[(size_t)y * w + x] = score_at(img + (size_t)y * step + x, step, minTh);
}

struct Cand { int x, y, score; };   // relative to (minBorderX, minBorderY)

// ComputeKeyPointsOctTree cell loop, src/ORBextractor.cc:1064-1157
void cell_candidates(const uint8_t *S, int cols, int rows, int iniTh, std::vector<Cand> &out)
{
    out.clear();
    const int minBX = kEdgeThreshold - 3, minBY = minBX;
    const int maxBX = cols - kEdgeThreshold + 3, maxBY = rows - kEdgeThreshold + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / kCellW), nRows = (int)(height / kCellW);
    const int wCell = (int)ceil(width / nCols), hCell = (int)ceil(height / nRows);
    std::vector<Cand> cell;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxX > maxBX) maxX = (float)maxBX;
            // detectable area of the sub-image [iniY,maxY) x [iniX,maxX): rows/cols 3..size-4
            const int y0 = (int)iniY + 3, y1 = (int)maxY - 3, x0 = (int)iniX + 3, x1 = (int)maxX - 3;
            cell.clear();
            bool anyIni = false;
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) {
                    int v = S[(size_t)y * cols + x];
                    if (v == 0) continue;
                    bool mx = true;
                    for (int dy = -1; dy <= 1 && mx; dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            if (!dx && !dy) continue;
                            int yy = y + dy, xx = x + dx;
                            int q = (yy >= y0 && yy < y1 && xx >= x0 && xx < x1) ? S[(size_t)yy * cols + xx] : 0;
                            if (!(v > q)) { mx = false; break; }
                        }
                    if (!mx) continue;
                    Cand c = {x - minBX, y - minBY, v};
                    cell.push_back(c);
                    if (v >= iniTh) anyIni = true;
                }
            for (size_t k = 0; k < cell.size(); k++)
                if (!anyIni || cell[k].score >= iniTh) out.push_back(cell[k]);
        }
    }
}

// DistributeOctTree + ExtractorNode::DivideNode, src/ORBextractor.cc:635-703, 706-1049,
// in array form.  Tie rule for the sort at :948 (pair<int,ExtractorNode*>): among equal
// point counts the later-created node is split first == smaller list position first.
struct Node { int x0, y0, x1, y1, start, cnt; };

struct Split { Node ch[4]; int n; };

Split split_node(const Node &nd, const std::vector<Cand> &c, std::vector<int> &perm, std::vector<int> &tmp)
{
    const int halfX = (int)ceil((float)(nd.x1 - nd.x0) / 2), halfY = (int)ceil((float)(nd.y1 - nd.y0) / 2);
    const int mx = nd.x0 + halfX, my = nd.y0 + halfY;
    int cnt[4] = {0, 0, 0, 0};
    for (int k = 0; k < nd.cnt; k++) {
        const Cand &p = c[(size_t)perm[(size_t)nd.start + k]];
        int q = ((float)p.x < (float)mx) ? (((float)p.y < (float)my) ? 0 : 2) : (((float)p.y < (float)my) ? 1 : 3);
        cnt[q]++;
    }
    int base[4], acc = nd.start;
    for (int q = 0; q < 4; q++) { base[q] = acc; acc += cnt[q]; }
    int fill[4] = {base[0], base[1], base[2], base[3]};
    for (int k = 0; k < nd.cnt; k++) {
        int id = perm[(size_t)nd.start + k];
        const Cand &p = c[(size_t)id];
        int q = ((float)p.x < (float)mx) ? (((float)p.y < (float)my) ? 0 : 2) : (((float)p.y < (float)my) ? 1 : 3);
        tmp[(size_t)fill[q]++] = id;
    }
    for (int k = 0; k < nd.cnt; k++) perm[(size_t)nd.start + k] = tmp[(size_t)nd.start + k];
    Node ch[4] = {{nd.x0, nd.y0, mx, my, base[0], cnt[0]},
                  {mx, nd.y0, nd.x1, my, base[1], cnt[1]},
                  {nd.x0, my, mx, nd.y1, base[2], cnt[2]},
                  {mx, my, nd.x1, nd.y1, base[3], cnt[3]}};
    Split s;
    s.n = 0;
    for (int q = 0; q < 4; q++)
        if (ch[q].cnt > 0) s.ch[s.n++] = ch[q];
    return s;
}

void octree(const std::vector<Cand> &c, int minX, int maxX, int minY, int maxY, int N, std::vector<Cand> &out)
{
    out.clear();
    const int M = (int)c.size();
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));   // :719
    const float hX = (float)(maxX - minX) / nIni;                          // :722
    std::vector<int> perm((size_t)M), tmp((size_t)M);
    std::vector<Node> list;
    {
        std::vector<std::vector<int> > bins((size_t)nIni);
        for (int i = 0; i < M; i++) bins[(size_t)((float)c[(size_t)i].x / hX)].push_back(i);   // :766
        int acc = 0;
        for (int i = 0; i < nIni; i++) {
            Node nd = {(int)(hX * (float)i), 0, (int)(hX * (float)(i + 1)), maxY - minY, acc, (int)bins[(size_t)i].size()};
            for (size_t k = 0; k < bins[(size_t)i].size(); k++) perm[(size_t)acc++] = bins[(size_t)i][k];
            if (nd.cnt > 0) list.push_back(nd);   // empty nodes erased, :779-780
        }
    }
    bool finish = false;
    while (!finish) {
        int prev = (int)list.size();
        std::vector<Node> created, singles;
        int nToExpand = 0;
        for (size_t i = 0; i < list.size(); i++) {        // full pass, :803-905
            if (list[i].cnt == 1) { singles.push_back(list[i]); continue; }
            Split s = split_node(list[i], c, perm, tmp);
            for (int k = 0; k < s.n; k++) {
                created.push_back(s.ch[k]);
                if (s.ch[k].cnt > 1) nToExpand++;
            }
        }
        std::reverse(created.begin(), created.end());     // push_front of each child
        list = created;
        list.insert(list.end(), singles.begin(), singles.end());
        if ((int)list.size() >= N || (int)list.size() == prev) {
            finish = true;
        } else if ((int)list.size() + nToExpand * 3 > N) {
            while (!finish) {                              // careful rounds, :934-1011
                prev = (int)list.size();
                std::vector<int> cand;
                for (size_t i = 0; i < list.size(); i++) if (list[i].cnt > 1) cand.push_back((int)i);
                std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return list[(size_t)a].cnt > list[(size_t)b].cnt; });
                std::vector<char> removed(list.size(), 0);
                std::vector<Node> cr;
                int size = (int)list.size();
                for (size_t k = 0; k < cand.size(); k++) {
                    Split s = split_node(list[(size_t)cand[k]], c, perm, tmp);
                    for (int q = 0; q < s.n; q++) cr.push_back(s.ch[q]);
                    removed[(size_t)cand[k]] = 1;
                    size += s.n - 1;
                    if (size >= N) break;
                }
                std::reverse(cr.begin(), cr.end());
                for (size_t i = 0; i < list.size(); i++) if (!removed[i]) cr.push_back(list[i]);
                list = cr;
                if ((int)list.size() >= N || (int)list.size() == prev) finish = true;
            }
        }
    }
    for (size_t i = 0; i < list.size(); i++) {            // best response per node, :1018-1048
        const Node &nd = list[i];
        int best = perm[(size_t)nd.start];
        for (int k = 1; k < nd.cnt; k++) {
            int id = perm[(size_t)nd.start + k];
            if (c[(size_t)id].score > c[(size_t)best].score) best = id;
        }
        out.push_back(c[(size_t)best]);
    }
}

// IC_Angle, src/ORBextractor.cc:108-161 (x,y integer level coordinates)
float ic_angle(const uint8_t *img, size_t step, int x, int y, const int *umax)
{
    int m01 = 0, m10 = 0;
    const uint8_t *center = img + (size_t)y * step + x;
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * center[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
        int vsum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int vp = center[u + v * (ptrdiff_t)step], vm = center[u - v * (ptrdiff_t)step];
            vsum += (vp - vm);
            m10 += u * (vp + vm);
        }
        m01 += v * vsum;
    }
    return op_fast_atan2((float)m01, (float)m10);
}

// computeOrbDescriptor, src/ORBextractor.cc:173-227
void descriptor(const uint8_t *blur, size_t step, int x, int y, float angleDeg, uint8_t *desc)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);   // :164
    float angle = angleDeg * factorPI, a, b;
    op_sincosf(angle, &b, &a);   // a = cos, b = sin (:181); pinned to libm by tests/test_sincos.py
    const uint8_t *center = blur + (size_t)y * step + x;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            const int8_t *p = kPattern + 4 * (8 * i + k);
            int r0 = op_round_f(p[0] * b + p[1] * a), c0 = op_round_f(p[0] * a - p[1] * b);
            int r1 = op_round_f(p[2] * b + p[3] * a), c1 = op_round_f(p[2] * a - p[3] * b);
            int t0 = center[(ptrdiff_t)r0 * (ptrdiff_t)step + c0], t1 = center[(ptrdiff_t)r1 * (ptrdiff_t)step + c1];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

struct Kp { float x, y, size, angle, response; int octave, class_id; };

}  // namespace

#define ORBO_API extern "C" __attribute__((visibility("default")))

ORBO_API void *orbo_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
{
    Oracle *o = new Oracle();
    o->nfeatures = nfeatures; o->scaleFactor = scaleFactor; o->nlevels = nlevels; o->iniTh = iniThFAST; o->minTh = minThFAST;
    build_tables(*o);
    return o;
}
ORBO_API void orbo_destroy(void *h) { delete (Oracle *)h; }

ORBO_API void orbo_tables(void *h, float *tables, int *quotas, int *umax)
{
    Oracle *o = (Oracle *)h;
    int nl = o->nlevels;
    for (int i = 0; i < nl; i++) {
        tables[i] = o->scale[i]; tables[nl + i] = o->invScale[i]; tables[2 * nl + i] = o->sigma2[i]; tables[3 * nl + i] = o->invSigma2[i];
        quotas[i] = o->quota[i];
    }
    for (int i = 0; i < 16; i++) umax[i] = o->umax[i];
}

ORBO_API void orbo_level_size(void *h, int W, int H, int level, int *w, int *hh) { level_size(*(Oracle *)h, W, H, level, w, hh); }

// level `level` (tight rows) from level `level-1` (tight rows) -- ComputePyramid :1696
ORBO_API void orbo_resize(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh)
{
    op_resize_linear_u8(src, sw, sh, (size_t)sw, dst, dw, dh, (size_t)dw);
}

ORBO_API void orbo_score_map(const uint8_t *img, int w, int hh, int stride, int minTh, uint8_t *out) { score_map(img, w, hh, (size_t)stride, minTh, out); }

// candidates of one level in vToDistributeKeys order, packed x | y<<12 | score<<24
ORBO_API int orbo_cell_candidates(const uint8_t *scores, int w, int hh, int iniTh, uint32_t *packed, int cap)
{
    std::vector<Cand> c;
    cell_candidates(scores, w, hh, iniTh, c);
    for (size_t i = 0; i < c.size() && (int)i < cap; i++) packed[i] = (uint32_t)c[i].x | ((uint32_t)c[i].y << 12) | ((uint32_t)c[i].score << 24);
    return (int)c.size();
}

ORBO_API int orbo_octree(const uint32_t *packed, int n, int minX, int maxX, int minY, int maxY, int N, uint32_t *out, int cap)
{
    std::vector<Cand> c((size_t)n), r;
    for (int i = 0; i < n; i++) { c[(size_t)i].x = packed[i] & 0xfff; c[(size_t)i].y = (packed[i] >> 12) & 0xfff; c[(size_t)i].score = packed[i] >> 24; }
    octree(c, minX, maxX, minY, maxY, N, r);
    for (size_t i = 0; i < r.size() && (int)i < cap; i++) out[i] = (uint32_t)r[i].x | ((uint32_t)r[i].y << 12) | ((uint32_t)r[i].score << 24);
    return (int)r.size();
}

ORBO_API float orbo_ic_angle(void *h, const uint8_t *img, int stride, int x, int y) { return ic_angle(img, (size_t)stride, x, y, ((Oracle *)h)->umax); }

ORBO_API void orbo_blur(const uint8_t *src, int w, int hh, int sstride, uint8_t *dst, int dstride, const uint16_t *taps)
{
    op_gauss7_u8(src, w, hh, (size_t)sstride, dst, (size_t)dstride, taps);
}

ORBO_API void orbo_descriptor(const uint8_t *blur, int stride, int x, int y, float angleDeg, uint8_t *desc) { descriptor(blur, (size_t)stride, x, y, angleDeg, desc); }

ORBO_API void orbo_sincos(const float *x, int n, float *sn, float *cs) { for (int i = 0; i < n; i++) op_sincosf(x[i], &sn[i], &cs[i]); }
// libm of this box, for tests/test_sincos.py
ORBO_API void orbo_libm_sincos(const float *x, int n, float *sn, float *cs) { for (int i = 0; i < n; i++) { sn[i] = sinf(x[i]); cs[i] = cosf(x[i]); } }
// exhaustive comparison over every float bit pattern in [lo_bits, hi_bits]; returns #mismatches
ORBO_API long orbo_sincos_exhaustive(uint32_t lo_bits, uint32_t hi_bits)
{
    long bad = 0;
    for (uint64_t u = lo_bits; u <= hi_bits; u++) {
        uint32_t b = (uint32_t)u;
        float f, s, c;
        memcpy(&f, &b, 4);
        op_sincosf(f, &s, &c);
        float ls = sinf(f), lc = cosf(f);
        if (memcmp(&s, &ls, 4) || memcmp(&c, &lc, 4)) bad++;
    }
    return bad;
}

// Full ORBextractor::operator(), src/ORBextractor.cc:1544-1668.  kps: 7 floats each as in
// oracle/ref_wrap.cc.  Optional stage outputs (any may be NULL): level_counts[nlevels].
ORBO_API int orbo_extract(void *h, const uint8_t *img, int W, int H, int stride, float *kps, uint8_t *desc, int cap, int *level_counts)
{
    Oracle *o = (Oracle *)h;
    if (!img || W <= 0 || H <= 0) return 0;
    const int nl = o->nlevels;
    std::vector<std::vector<uint8_t> > pyr((size_t)nl);
    std::vector<int> lw((size_t)nl), lh((size_t)nl);
    for (int l = 0; l < nl; l++) {
        level_size(*o, W, H, l, &lw[(size_t)l], &lh[(size_t)l]);
        pyr[(size_t)l].resize((size_t)lw[(size_t)l] * lh[(size_t)l]);
        if (l == 0) for (int y = 0; y < H; y++) memcpy(&pyr[0][(size_t)y * W], img + (size_t)y * stride, (size_t)W);
        else op_resize_linear_u8(pyr[(size_t)l - 1].data(), lw[(size_t)l - 1], lh[(size_t)l - 1], (size_t)lw[(size_t)l - 1],
                                 pyr[(size_t)l].data(), lw[(size_t)l], lh[(size_t)l], (size_t)lw[(size_t)l]);
    }
    int n = 0;
    std::vector<uint8_t> S, blur;
    std::vector<Cand> cand, sel;
    for (int l = 0; l < nl; l++) {
        const int w = lw[(size_t)l], hh = lh[(size_t)l];
        const uint8_t *im = pyr[(size_t)l].data();
        S.resize((size_t)w * hh);
        score_map(im, w, hh, (size_t)w, o->minTh, S.data());
        cell_candidates(S.data(), w, hh, o->iniTh, cand);
        const int minB = kEdgeThreshold - 3, maxBX = w - kEdgeThreshold + 3, maxBY = hh - kEdgeThreshold + 3;
        octree(cand, minB, maxBX, minB, maxBY, o->quota[(size_t)l], sel);
        if (level_counts) level_counts[l] = (int)sel.size();
        if (sel.empty()) continue;
        blur.resize((size_t)w * hh);
        op_gauss7_u8(im, w, hh, (size_t)w, blur.data(), (size_t)w, NULL);
        const int scaledPatch = (int)(kPatchSize * o->scale[(size_t)l]);   // :1175
        const float sc = o->scale[(size_t)l];
        for (size_t i = 0; i < sel.size(); i++, n++) {
            if (n >= cap) continue;
            const int x = sel[i].x + minB, y = sel[i].y + minB;
            const float ang = ic_angle(im, (size_t)w, x, y, o->umax);
            descriptor(blur.data(), (size_t)w, x, y, ang, desc + 32 * (size_t)n);
            float *k = kps + 7 * (size_t)n;
            k[0] = l ? (float)x * sc : (float)x;   // pt *= scale for level != 0 (:1651-1660)
            k[1] = l ? (float)y * sc : (float)y;
            k[2] = (float)scaledPatch; k[3] = ang; k[4] = (float)sel[i].score; k[5] = (float)l; k[6] = -1.f;
        }
    }
    return n;
}

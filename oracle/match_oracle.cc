// oracle/match_oracle.cc -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement, on flat arrays, of the Hamming matchers of the reference:
//   ORBmatcher::DescriptorDistance            src/ORBmatcher.cc:1913-1933
//   ORBmatcher::SearchByBoW(KeyFrame*,Frame&) src/ORBmatcher.cc:230-382
//   ORBmatcher::SearchByBoW(KeyFrame*,KeyFrame*) src/ORBmatcher.cc:656-799
//   ORBmatcher::ComputeThreeMaxima            src/ORBmatcher.cc:1866-1908
//   Frame::ComputeStereoMatches, Hamming stage src/Frame.cc:1041-1216
//   Frame::ComputeStereoMatches, complete      src/Frame.cc:1026-1420
// PINNED by the reference itself: oracle/_ref/liborbslam.so is the unmodified
// src/ORBmatcher.cc + Frame.cc + KeyFrame.cc + MapPoint.cc + DBoW2 compiled against
// oracle/cvshim, and tests/test_refslam.py checks every function here against it (real
// KeyFrame / Frame / MapPoint objects) on seeded inputs; tests/golden/ holds its outputs.
// What is kept literally: the bit-hack distance, scan order (ascending node id, ascending
// feature index inside a node), strict '<' updates (first minimum wins), the skip of
// already matched F features, thresholds, ratio test in float, the 30-bin rotation
// histogram with round() and the three-maxima pruning.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#define MO_API extern "C" __attribute__((visibility("default")))

namespace {

const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;   // src/ORBmatcher.cc:49-51

// src/ORBmatcher.cc:1913-1933 (8 x int32 bit-hack popcount)
int descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// src/ORBmatcher.cc:1866-1908
void three_maxima(const std::vector<int> *histo, int L, int &ind1, int &ind2, int &ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

typedef std::map<int, std::vector<unsigned> > FeatVec;   // DBoW2::FeatureVector: node id -> ascending feature indices

FeatVec make_featvec(const int32_t *groups, int n)
{
    FeatVec fv;
    for (int i = 0; i < n; i++) {
        const int g = groups ? groups[i] : 0;
        if (g >= 0) fv[g].push_back((unsigned)i);   // addFeature keeps ascending order; negative = not filed (word weight 0)
    }
    return fv;
}

}  // namespace

MO_API int mo_descriptor_distance(const uint8_t *a, const uint8_t *b) { return descriptor_distance(a, b); }

// mode 0: SearchByBoW(KeyFrame*, Frame&):  match_out has nB entries (index of the KF feature or -1)
// mode 1: SearchByBoW(KeyFrame*, KeyFrame*): match_out has nA entries (index of the KF2 feature or -1)
// validA/validB: 1 where the feature has a non-bad MapPoint (NULL = all valid; validB only used in mode 1).
MO_API int mo_search_by_bow(int mode, const uint8_t *descA, const float *angleA, const int32_t *groupA, const uint8_t *validA, int nA,
                            const uint8_t *descB, const float *angleB, const int32_t *groupB, const uint8_t *validB, int nB,
                            float nnratio, int checkOri, int32_t *match_out)
{
    const int nOut = mode == 0 ? nB : nA;
    for (int i = 0; i < nOut; i++) match_out[i] = -1;
    std::vector<char> matchedB((size_t)nB, 0);
    FeatVec fA = make_featvec(groupA, nA), fB = make_featvec(groupB, nB);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = HISTO_LENGTH / 360.0f;
    FeatVec::const_iterator ia = fA.begin(), ib = fB.begin();
    while (ia != fA.end() && ib != fB.end()) {
        if (ia->first == ib->first) {
            for (size_t k = 0; k < ia->second.size(); k++) {
                const unsigned idxA = ia->second[k];
                if (validA && !validA[idxA]) continue;
                int best1 = 256, bestIdx = -1, best2 = 256;
                for (size_t m = 0; m < ib->second.size(); m++) {
                    const unsigned idxB = ib->second[m];
                    if (matchedB[idxB]) continue;                         // :288 / :717
                    if (mode == 1 && validB && !validB[idxB]) continue;   // :717-721
                    const int dist = descriptor_distance(descA + 32 * (size_t)idxA, descB + 32 * (size_t)idxB);
                    if (dist < best1) { best2 = best1; best1 = dist; bestIdx = (int)idxB; }
                    else if (dist < best2) { best2 = dist; }
                }
                const bool pass = mode == 0 ? (best1 <= TH_LOW) : (best1 < TH_LOW);   // :308 vs :741
                if (pass && (float)best1 < nnratio * (float)best2) {
                    matchedB[(size_t)bestIdx] = 1;
                    const int slot = mode == 0 ? bestIdx : (int)idxA;
                    match_out[slot] = mode == 0 ? (int)idxA : bestIdx;
                    if (checkOri) {
                        float rot = angleA[idxA] - angleB[bestIdx];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(slot);
                    }
                    nmatches++;
                }
            }
            ++ia; ++ib;
        } else if (ia->first < ib->first) ia = fA.lower_bound(ib->first);
        else ib = fB.lower_bound(ia->first);
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { match_out[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// Frame::ComputeStereoMatches up to the ORB-distance decision (src/Frame.cc:1041-1216):
// kp arrays are 7 floats per keypoint {x,y,size,angle,response,octave,class_id};
// best_dist[iL] = TH_HIGH when no candidate beat it, best_idx[iL] = winning right index (0 if none).
MO_API void mo_stereo_hamming(const float *kpL, const uint8_t *descL, int nL, const float *kpR, const uint8_t *descR, int nR,
                              const float *scaleFactors, int nRows, float maxD, int32_t *best_dist, int32_t *best_idx)
{
    std::vector<std::vector<size_t> > vRowIndices((size_t)nRows);
    for (int iR = 0; iR < nR; iR++) {
        const float kpY = kpR[7 * iR + 1];
        const float r = 2.0f * scaleFactors[(int)kpR[7 * iR + 5]];
        const int maxr = (int)ceil(kpY + r), minr = (int)floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[(size_t)yi].push_back((size_t)iR);
    }
    const float minD = 0;
    for (int iL = 0; iL < nL; iL++) {
        best_dist[iL] = TH_HIGH;
        best_idx[iL] = 0;
        const int levelL = (int)kpL[7 * iL + 5];
        const float vL = kpL[7 * iL + 1], uL = kpL[7 * iL];
        const std::vector<size_t> &cands = vRowIndices[(size_t)vL];
        if (cands.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        for (size_t iC = 0; iC < cands.size(); iC++) {
            const size_t iR = cands[iC];
            const int octR = (int)kpR[7 * iR + 5];
            if (octR < levelL - 1 || octR > levelL + 1) continue;
            const float uR = kpR[7 * iR];
            if (uR >= minU && uR <= maxU) {
                const int dist = descriptor_distance(descL + 32 * (size_t)iL, descR + 32 * iR);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        best_dist[iL] = bestDist;
        best_idx[iL] = (int32_t)bestIdxR;
    }
}


// Frame::ComputeStereoMatches, complete (src/Frame.cc:1026-1420): Hamming stage as above,
// then the 11x11 SAD search over +-5 px on the keypoint's pyramid level (:1224-1306), the
// parabola sub-pixel fit (:1330-1347), the disparity gate (:1355-1378) and the
// median * 1.5 * 1.4 outlier cut (:1387-1416).  pyrL/pyrR: the nlevels level images (tight
// rows, sizes lw/lh) concatenated.  mb is what Frame::mb holds when the function runs: 0 in
// the stereo constructor of this fork (src/Frame.cc:125) => maxD = +inf.
// Float behaviour kept: round() on float products, SAD distances are exact integers in
// float, `bestDist = dist` truncates float->int, deltaR/bestuR in float, the 0.01 clamp in
// double (`uL-0.01`).  An empty vDistIdx is undefined behaviour in the reference
// (vDistIdx[0] of an empty vector, :1388); here nothing is cut in that case.
MO_API void mo_compute_stereo_matches(const float *kpL, const uint8_t *descL, int nL, const float *kpR, const uint8_t *descR, int nR,
                                      const uint8_t *pyrL, const uint8_t *pyrR, const int *lw, const int *lh, int nlevels,
                                      const float *scaleFactors, const float *invScaleFactors, float mbf, float mb,
                                      float *uRight, float *depth, int32_t *sad_out)
{
    const int nRows = lh[0];
    std::vector<int32_t> bd((size_t)(nL > 0 ? nL : 1)), bi((size_t)(nL > 0 ? nL : 1));
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    mo_stereo_hamming(kpL, descL, nL, kpR, descR, nR, scaleFactors, nRows, maxD, bd.data(), bi.data());
    std::vector<size_t> off((size_t)nlevels);
    size_t o = 0;
    for (int l = 0; l < nlevels; l++) { off[(size_t)l] = o; o += (size_t)lw[l] * lh[l]; }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    std::vector<std::pair<int, int> > vDistIdx;
    for (int iL = 0; iL < nL; iL++) {
        uRight[iL] = -1.0f;
        depth[iL] = -1.0f;
        if (sad_out) sad_out[iL] = -1;
        // the reference `continue`s before the Hamming loop for an empty row / maxU < 0: bd stays TH_HIGH
        if (!(bd[(size_t)iL] < thOrbDist)) continue;
        const int oct = (int)kpL[7 * iL + 5];
        const float uL = kpL[7 * iL], vL = kpL[7 * iL + 1];
        const float uR0 = kpR[7 * (size_t)bi[(size_t)iL]];
        const float scaleFactor = invScaleFactors[oct];
        const float scaleduL = roundf(uL * scaleFactor);
        const float scaledvL = roundf(vL * scaleFactor);
        const float scaleduR0 = roundf(uR0 * scaleFactor);
        const int w = 5, L = 5;
        const int W = lw[oct];
        const uint8_t *imL = pyrL + off[(size_t)oct], *imR = pyrR + off[(size_t)oct];
        const int cy = (int)scaledvL, cxl = (int)scaleduL;
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= W) continue;
        const int centerL = imL[(size_t)cy * W + cxl];
        int bestDist = 0x7fffffff, bestincR = 0;
        float vDists[2 * 5 + 1];
        for (int incR = -L; incR <= L; incR++) {
            const int cxr = (int)(scaleduR0 + incR);   // colRange(scaleduR0+incR-w, ...) truncates the float sum
            const int centerR = imR[(size_t)cy * W + cxr];
            int sad = 0;
            for (int dy = -w; dy <= w; dy++)
                for (int dx = -w; dx <= w; dx++) {
                    int a = (int)imL[(size_t)(cy + dy) * W + cxl + dx] - centerL;
                    int b = (int)imR[(size_t)(cy + dy) * W + cxr + dx] - centerR;
                    sad += a > b ? a - b : b - a;
                }
            const float dist = (float)sad;
            if (dist < bestDist) { bestDist = (int)dist; bestincR = incR; }
            vDists[L + incR] = dist;
        }
        if (bestincR == -L || bestincR == L) continue;
        const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
        const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
        if (deltaR < -1 || deltaR > 1) continue;
        float bestuR = scaleFactors[oct] * ((float)scaleduR0 + (float)bestincR + deltaR);
        float disparity = (uL - bestuR);
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
            depth[iL] = mbf / disparity;
            uRight[iL] = bestuR;
            if (sad_out) sad_out[iL] = bestDist;
            vDistIdx.push_back(std::pair<int, int>(bestDist, iL));
        }
    }
    if (vDistIdx.empty()) return;
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[(size_t)i].first < thDist) break;
        uRight[vDistIdx[(size_t)i].second] = -1;
        depth[vDistIdx[(size_t)i].second] = -1;
    }
}

// ---------------------------------------------------------------------------------------
// DBoW2 TemplatedVocabulary::transform on a flat tree (Thirdparty/DBoW2/DBoW2/
// TemplatedVocabulary.h:1127-1262, FORB::distance FORB.cpp:81-101).  Node 0 is the root;
// parent[i] < i; the children of a node are its child ids in ascending order (what
// loadFromTextFile builds, :1378-1420).  From the root: pick the child of minimum Hamming
// distance, FIRST minimum wins (strict '<', :1219-1229), until a leaf; node_out = the node
// reached at depth L - levelsup (0 when that is <= 0, unchanged... the reference leaves *nid
// untouched when the leaf is reached before that depth: the wrapper initialises it to 0).
// PINNED against the compiled reference (oracle/_ref/liborbslam.so) in tests/test_bow_transform.py.
// ---------------------------------------------------------------------------------------
MO_API void mo_voc_transform(int num_nodes, const int32_t *parent, const uint8_t *is_leaf, const uint8_t *node_desc, const double *node_weight,
                             const int32_t *node_word, int L, const uint8_t *desc, int n, int levelsup, int32_t *word, int32_t *node,
                             double *weight)
{
    std::vector<std::vector<int> > children((size_t)num_nodes);
    for (int i = 1; i < num_nodes; i++) children[(size_t)parent[i]].push_back(i);
    const int nid_level = L - levelsup;
    for (int f = 0; f < n; f++) {
        const uint8_t *d = desc + 32 * (size_t)f;
        int final_id = 0, current_level = 0, nid = 0;
        do {
            ++current_level;
            const std::vector<int> &ch = children[(size_t)final_id];
            final_id = ch[0];
            int best = descriptor_distance(d, node_desc + 32 * (size_t)final_id);
            for (size_t c = 1; c < ch.size(); c++) {
                const int dd = descriptor_distance(d, node_desc + 32 * (size_t)ch[c]);
                if (dd < best) { best = dd; final_id = ch[c]; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (!is_leaf[final_id]);
        word[f] = node_word[final_id];
        weight[f] = node_weight[final_id];
        node[f] = nid;
    }
}

// ---------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)
// (src/ORBmatcher.cc:70-175) with Frame::GetFeaturesInArea (src/Frame.cc:741-850) and
// Frame::PosInGrid / AssignFeaturesToGrid (:461-491, 853-877) on flat arrays.
//   frame side : mvKeysUn (7 floats each), mDescriptors, mvuRight, occupied[idx] = 1 when
//                F.mvpMapPoints[idx] holds a MapPoint with Observations()>0 at entry, grid bounds
//   point side : what Frame::isInFrustum left in each MapPoint (mTrackProjX/Y/XR, mnTrackScaleLevel,
//                mTrackViewCos, mbTrackInView && !isBad()), has_obs = Observations()>0, descriptor
// assigned[idx] = index of the MapPoint written into F.mvpMapPoints[idx] by this call, else -1.
// PINNED against the compiled reference in tests/test_projection.py.
// ---------------------------------------------------------------------------------------
namespace {
const int GRID_COLS = 64, GRID_ROWS = 48;   // include/Frame.h:55-60
}

MO_API int mo_search_by_projection(const float *kpUn, const uint8_t *desc, const float *uRight, const uint8_t *occupied_in, int n, float minX, float minY,
                                   float gridWInv, float gridHInv, const float *scaleFactors, const float *projX, const float *projY,
                                   const float *projXR, const int32_t *level, const float *viewCos, const uint8_t *inView, const uint8_t *hasObs,
                                   const uint8_t *mpDesc, int m, float th, float nnratio, int32_t *assigned)
{
    // AssignFeaturesToGrid: cell lists hold feature indices in ascending order
    std::vector<std::vector<size_t> > grid((size_t)GRID_COLS * GRID_ROWS);
    for (int i = 0; i < n; i++) {
        const int posX = (int)roundf((kpUn[7 * i] - minX) * gridWInv), posY = (int)roundf((kpUn[7 * i + 1] - minY) * gridHInv);
        if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
        grid[(size_t)posX * GRID_ROWS + posY].push_back((size_t)i);
    }
    std::vector<uint8_t> occupied(occupied_in, occupied_in + n);
    for (int i = 0; i < n; i++) assigned[i] = -1;
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (int iMP = 0; iMP < m; iMP++) {
        if (!inView[iMP]) continue;
        const int nPredictedLevel = level[iMP];
        float r = viewCos[iMP] > 0.998 ? 2.5f : 4.0f;   // RadiusByViewingCos, :178-185
        if (bFactor) r *= th;
        const float x = projX[iMP], y = projY[iMP], rr = r * scaleFactors[nPredictedLevel];
        const int minLevel = nPredictedLevel - 1, maxLevel = nPredictedLevel;
        // GetFeaturesInArea
        std::vector<size_t> vIndices;
        do {
            const int nMinCellX = std::max(0, (int)floor((x - minX - rr) * gridWInv));
            if (nMinCellX >= GRID_COLS) break;
            const int nMaxCellX = std::min(GRID_COLS - 1, (int)ceil((x - minX + rr) * gridWInv));
            if (nMaxCellX < 0) break;
            const int nMinCellY = std::max(0, (int)floor((y - minY - rr) * gridHInv));
            if (nMinCellY >= GRID_ROWS) break;
            const int nMaxCellY = std::min(GRID_ROWS - 1, (int)ceil((y - minY + rr) * gridHInv));
            if (nMaxCellY < 0) break;
            const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
                for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                    const std::vector<size_t> &vCell = grid[(size_t)ix * GRID_ROWS + iy];
                    for (size_t j = 0; j < vCell.size(); j++) {
                        const float *kp = kpUn + 7 * vCell[j];
                        const int oct = (int)kp[5];
                        if (bCheckLevels) {
                            if (oct < minLevel) continue;
                            if (maxLevel >= 0 && oct > maxLevel) continue;
                        }
                        const float distx = kp[0] - x, disty = kp[1] - y;
                        if (fabs(distx) < rr && fabs(disty) < rr) vIndices.push_back(vCell[j]);
                    }
                }
        } while (0);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t v = 0; v < vIndices.size(); v++) {
            const size_t idx = vIndices[v];
            if (occupied[idx]) continue;                                   // mvpMapPoints[idx] && Observations()>0, :110-112
            if (uRight[idx] > 0) {
                const float er = fabs(projXR[iMP] - uRight[idx]);
                if (er > rr) continue;
            }
            const int dist = descriptor_distance(mpDesc + 32 * (size_t)iMP, desc + 32 * idx);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel;
                bestLevel = (int)kpUn[7 * idx + 5]; bestIdx = (int)idx;
            } else if (dist < bestDist2) {
                bestLevel2 = (int)kpUn[7 * idx + 5]; bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            assigned[bestIdx] = iMP;                                       // F.mvpMapPoints[bestIdx]=pMP
            occupied[(size_t)bestIdx] = hasObs[iMP] ? 1 : 0;               // what the check at :110-112 will see from now on
            nmatches++;
        }
    }
    return nmatches;
}

// ---------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th,
// const bool bMono) (src/ORBmatcher.cc:1569-1728), the matcher of Tracking::TrackWithMotionModel.
// Last-frame side per feature i: valid[i] = mvpMapPoints[i] != NULL && !mvbOutlier[i], the
// MapPoint's world position, descriptor and Observations()>0, mvKeys[i].octave, mvKeysUn[i].angle.
// Projection arithmetic in float, left to right, no contraction (the small-matrix path of
// OpenCV's gemm as restated in oracle/cvshim).  assigned[i2] = last-frame feature whose MapPoint
// was written into CurrentFrame.mvpMapPoints[i2]; -1 = untouched, -2 = written and then cleared by the
// rotation pruning.
// ---------------------------------------------------------------------------------------
MO_API int mo_search_by_projection_last(const float *kpUn, const uint8_t *desc, const float *uRight, const uint8_t *occupied_in, int n, float minX,
                                        float minY, float maxX, float maxY, float gridWInv, float gridHInv, const float *scaleFactors,
                                        const float *TcwCur, const float *TcwLast, float fx, float fy, float cx, float cy, float mbf, float mb,
                                        const uint8_t *lastValid, const float *lastPos, const uint8_t *lastDesc, const uint8_t *lastHasObs,
                                        const int32_t *lastOctave, const float *lastAngle, int nLast, float th, int bMono, int checkOri,
                                        int32_t *assigned)
{
    std::vector<std::vector<size_t> > grid((size_t)GRID_COLS * GRID_ROWS);
    for (int i = 0; i < n; i++) {
        const int posX = (int)roundf((kpUn[7 * i] - minX) * gridWInv), posY = (int)roundf((kpUn[7 * i + 1] - minY) * gridHInv);
        if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
        grid[(size_t)posX * GRID_ROWS + posY].push_back((size_t)i);
    }
    std::vector<uint8_t> occupied(occupied_in, occupied_in + n);
    for (int i = 0; i < n; i++) assigned[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = HISTO_LENGTH / 360.0f;
    // Rcw, tcw of the current frame; twc = -Rcw^T * tcw; tlc = Rlw*twc + tlw  (:1586-1596)
    const float *Rc = TcwCur, *Rl = TcwLast;   // row-major 4x4
    float twc[3], tlc[3];
    for (int r = 0; r < 3; r++) {
        float s = (-Rc[0 * 4 + r]) * Rc[0 * 4 + 3];
        s = s + (-Rc[1 * 4 + r]) * Rc[1 * 4 + 3];
        s = s + (-Rc[2 * 4 + r]) * Rc[2 * 4 + 3];
        twc[r] = s;
    }
    for (int r = 0; r < 3; r++) {
        float s = Rl[r * 4 + 0] * twc[0];
        s = s + Rl[r * 4 + 1] * twc[1];
        s = s + Rl[r * 4 + 2] * twc[2];
        tlc[r] = s + Rl[r * 4 + 3];
    }
    const bool bForward = tlc[2] > mb && !bMono, bBackward = -tlc[2] > mb && !bMono;
    for (int i = 0; i < nLast; i++) {
        if (!lastValid[i]) continue;
        const float *X = lastPos + 3 * (size_t)i;
        float x3Dc[3];
        for (int r = 0; r < 3; r++) {
            float s = Rc[r * 4 + 0] * X[0];
            s = s + Rc[r * 4 + 1] * X[1];
            s = s + Rc[r * 4 + 2] * X[2];
            x3Dc[r] = s + Rc[r * 4 + 3];
        }
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = 1.0 / x3Dc[2];
        if (invzc < 0) continue;
        float u = fx * xc * invzc + cx, v = fy * yc * invzc + cy;
        if (u < minX || u > maxX) continue;
        if (v < minY || v > maxY) continue;
        const int nLastOctave = lastOctave[i];
        const float radius = th * scaleFactors[nLastOctave];
        int minLevel, maxLevel;
        if (bForward) { minLevel = nLastOctave; maxLevel = -1; }
        else if (bBackward) { minLevel = 0; maxLevel = nLastOctave; }
        else { minLevel = nLastOctave - 1; maxLevel = nLastOctave + 1; }
        std::vector<size_t> vIndices2;
        do {
            const int nMinCellX = std::max(0, (int)floor((u - minX - radius) * gridWInv));
            if (nMinCellX >= GRID_COLS) break;
            const int nMaxCellX = std::min(GRID_COLS - 1, (int)ceil((u - minX + radius) * gridWInv));
            if (nMaxCellX < 0) break;
            const int nMinCellY = std::max(0, (int)floor((v - minY - radius) * gridHInv));
            if (nMinCellY >= GRID_ROWS) break;
            const int nMaxCellY = std::min(GRID_ROWS - 1, (int)ceil((v - minY + radius) * gridHInv));
            if (nMaxCellY < 0) break;
            const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
                for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                    const std::vector<size_t> &vCell = grid[(size_t)ix * GRID_ROWS + iy];
                    for (size_t j = 0; j < vCell.size(); j++) {
                        const float *kp = kpUn + 7 * vCell[j];
                        const int oct = (int)kp[5];
                        if (bCheckLevels) {
                            if (oct < minLevel) continue;
                            if (maxLevel >= 0 && oct > maxLevel) continue;
                        }
                        const float distx = kp[0] - u, disty = kp[1] - v;
                        if (fabs(distx) < radius && fabs(disty) < radius) vIndices2.push_back(vCell[j]);
                    }
                }
        } while (0);
        if (vIndices2.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            const size_t i2 = vIndices2[k];
            if (occupied[i2]) continue;
            if (uRight[i2] > 0) {
                const float ur = u - mbf * invzc;
                const float er = fabs(ur - uRight[i2]);
                if (er > radius) continue;
            }
            const int dist = descriptor_distance(lastDesc + 32 * (size_t)i, desc + 32 * i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestDist <= TH_HIGH) {
            assigned[bestIdx2] = i;
            occupied[(size_t)bestIdx2] = lastHasObs[i] ? 1 : 0;
            nmatches++;
            if (checkOri) {
                float rot = lastAngle[i] - kpUn[7 * bestIdx2 + 3];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { assigned[rotHist[i][j]] = -2; nmatches--; }   // -2: assigned, then set to NULL (:1718)
        }
    }
    return nmatches;
}

// ---------------------------------------------------------------------------------------
// Frame::UndistortKeyPoints (src/Frame.cc:899-947), Frame::ComputeImageBounds (:950-1004),
// the grid constants of the constructor (src/Frame.cc:326-327) and
// Frame::AssignFeaturesToGrid / PosInGrid (:460-491, 868-878) on flat arrays.
// cam = fx, fy, cx, cy; dist = ndist (4 or 5) coefficients k1 k2 p1 p2 [k3] as stored in
// mDistCoef (float).  bounds = mnMinX, mnMaxX, mnMinY, mnMaxY; grid as CSR over
// cell = x*48 + y with the indices of a cell in ascending order (= push_back order).
// cv::undistortPoints is prims.h's op_undistort_points (parity unpinned at that level).
// ---------------------------------------------------------------------------------------
#include "prims.h"

namespace {

void undistort(const float *src, float *dst, int n, const float *cam, const float *dist, int ndist)
{
    double K[9] = {cam[0], 0, cam[2], 0, cam[1], cam[3], 0, 0, 1}, k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ndist; i++) k[i] = dist[i];
    op_undistort_points(src, dst, n, K, k, K, 5);
}
}  // namespace

MO_API void mo_frame_finish(const float *kp, int n, const float *cam, const float *dist, int ndist, int cols, int rows, float *kpUn,
                            float *bounds, float *gridInv, int32_t *gridOff, int32_t *gridIdx)
{
    const bool distorted = dist[0] != 0.0f;   // :901, :953
    std::vector<float> xy((size_t)2 * std::max(n, 1));
    for (int i = 0; i < n; i++) { xy[2 * i] = kp[7 * i]; xy[2 * i + 1] = kp[7 * i + 1]; }
    if (distorted) undistort(xy.data(), xy.data(), n, cam, dist, ndist);
    for (int i = 0; i < n; i++) {
        memcpy(kpUn + 7 * (size_t)i, kp + 7 * (size_t)i, 7 * sizeof(float));
        kpUn[7 * (size_t)i] = xy[2 * i]; kpUn[7 * (size_t)i + 1] = xy[2 * i + 1];
    }
    if (distorted) {
        float c[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
        undistort(c, c, 4, cam, dist, ndist);
        bounds[0] = std::min(c[0], c[4]); bounds[1] = std::max(c[2], c[6]);
        bounds[2] = std::min(c[1], c[3]); bounds[3] = std::max(c[5], c[7]);
    } else {
        bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows;
    }
    gridInv[0] = (float)GRID_COLS / (float)(bounds[1] - bounds[0]);
    gridInv[1] = (float)GRID_ROWS / (float)(bounds[3] - bounds[2]);
    std::vector<std::vector<int32_t> > grid((size_t)GRID_COLS * GRID_ROWS);
    for (int i = 0; i < n; i++) {
        const int px = (int)round((kpUn[7 * (size_t)i] - bounds[0]) * gridInv[0]);
        const int py = (int)round((kpUn[7 * (size_t)i + 1] - bounds[2]) * gridInv[1]);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
        grid[(size_t)px * GRID_ROWS + py].push_back(i);
    }
    int o = 0;
    for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) {
        gridOff[c] = o;
        for (size_t t = 0; t < grid[(size_t)c].size(); t++) gridIdx[o++] = grid[(size_t)c][t];
    }
    gridOff[GRID_COLS * GRID_ROWS] = o;
}

// ---------------------------------------------------------------------------------------
// ORBmatcher::SearchForTriangulation, src/ORBmatcher.cc:810-1017 (+ CheckDistEpipolarLine :188-227).
// Flat arrays; ex, ey (:817-826) are inputs.  hasMp*: feature holds a MapPoint; uRight*: mvuRight.
// sf / sigma2: pKF2->mvScaleFactors / mvLevelSigma2.  matches12[i] = KF2 feature or -1.
// ---------------------------------------------------------------------------------------
MO_API int mo_search_for_triangulation(const float *kpA, const uint8_t *descA, const int32_t *groupA, const uint8_t *hasMpA, const float *uRightA, int nA,
                                       const float *kpB, const uint8_t *descB, const int32_t *groupB, const uint8_t *hasMpB, const float *uRightB, int nB,
                                       const float *F12, float ex, float ey, const float *sf, const float *sigma2, int onlyStereo, int checkOri,
                                       int32_t *matches12)
{
    FeatVec fv1 = make_featvec(groupA, nA), fv2 = make_featvec(groupB, nB);
    int nmatches = 0;
    std::vector<char> matched2((size_t)nB, 0);
    for (int i = 0; i < nA; i++) matches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = HISTO_LENGTH / 360.0f;
    FeatVec::const_iterator f1 = fv1.begin(), f2 = fv2.begin();
    while (f1 != fv1.end() && f2 != fv2.end()) {
        if (f1->first == f2->first) {
            for (size_t i1 = 0; i1 < f1->second.size(); i1++) {
                const int idx1 = f1->second[i1];
                if (hasMpA && hasMpA[idx1]) continue;                                        // :845-848
                const bool st1 = uRightA[idx1] >= 0;
                if (onlyStereo && !st1) continue;                                            // :852-855
                const float *k1 = kpA + 7 * (size_t)idx1;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (size_t i2 = 0; i2 < f2->second.size(); i2++) {
                    const int idx2 = f2->second[i2];
                    if (matched2[(size_t)idx2] || (hasMpB && hasMpB[idx2])) continue;        // :867-870
                    const bool st2 = uRightB[idx2] >= 0;
                    if (onlyStereo && !st2) continue;
                    const int dist = descriptor_distance(descA + 32 * (size_t)idx1, descB + 32 * (size_t)idx2);
                    if (dist > TH_LOW || dist > bestDist) continue;                          // :880
                    const float *k2 = kpB + 7 * (size_t)idx2;
                    const int oct2 = (int)k2[5];
                    if (!st1 && !st2) {                                                      // :888-895
                        const float distex = ex - k2[0], distey = ey - k2[1];
                        if (distex * distex + distey * distey < 100 * sf[oct2]) continue;
                    }
                    // CheckDistEpipolarLine, :188-227
                    const float a = k1[0] * F12[0] + k1[1] * F12[3] + F12[6];
                    const float b = k1[0] * F12[1] + k1[1] * F12[4] + F12[7];
                    const float c = k1[0] * F12[2] + k1[1] * F12[5] + F12[8];
                    const float num = a * k2[0] + b * k2[1] + c;
                    const float den = a * a + b * b;
                    if (den == 0) continue;
                    const float dsqr = num * num / den;
                    if (dsqr < 3.84 * sigma2[oct2]) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    matches12[idx1] = bestIdx2;
                    matched2[(size_t)bestIdx2] = 1;
                    nmatches++;
                    if (checkOri) {
                        float rot = k1[3] - kpB[7 * (size_t)bestIdx2 + 3];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(idx1);
                    }
                }
            }
            ++f1; ++f2;
        } else if (f1->first < f2->first) f1 = fv1.lower_bound(f2->first);
        else f2 = fv2.lower_bound(f1->first);
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { matches12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ---------------------------------------------------------------------------------------
// ORBmatcher::Fuse (both overloads), steps 2-3 (src/ORBmatcher.cc:1093-1146, 1258-1276) with
// KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:752-796) on flat arrays: per map point the feature of
// minimum distance (strict '<': the first minimum in cell-major order wins) among the features in the
// window that pass the level gate and, with chi2Gate, the reprojection gate.  bestDist 256 when none.
// ---------------------------------------------------------------------------------------
MO_API void mo_fuse_best(const float *kpUn, const uint8_t *desc, const float *uRight, int n, float minX, float minY, float kfMinX, float kfMinY,
                         float gwInv, float ghInv,
                         const float *invSigma2, const float *pu, const float *pv, const float *pur, const int32_t *plevel, const float *pradius,
                         const uint8_t *pactive, const uint8_t *pdesc, int m, int chi2Gate, int32_t *bestIdx, int32_t *bestDist)
{
    std::vector<std::vector<int> > grid((size_t)GRID_COLS * GRID_ROWS);   // as AssignFeaturesToGrid files it
    for (int i = 0; i < n; i++) {
        const int px = (int)round((kpUn[7 * (size_t)i] - minX) * gwInv), py = (int)round((kpUn[7 * (size_t)i + 1] - minY) * ghInv);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
        grid[(size_t)px * GRID_ROWS + py].push_back(i);
    }
    for (int p = 0; p < m; p++) {
        bestIdx[p] = -1; bestDist[p] = 256;
        if (pactive && !pactive[p]) continue;
        const float x = pu[p], y = pv[p], r = pradius[p];
        const int lvl = plevel[p];
        // KeyFrame::GetFeaturesInArea with the KeyFrame's int bounds (kfMinX = (float)(int)mnMinX), src/KeyFrame.cc:760-775
        const int cx0 = std::max(0, (int)floor((x - kfMinX - r) * gwInv));
        if (cx0 >= GRID_COLS) continue;
        const int cx1 = std::min(GRID_COLS - 1, (int)ceil((x - kfMinX + r) * gwInv));
        if (cx1 < 0) continue;
        const int cy0 = std::max(0, (int)floor((y - kfMinY - r) * ghInv));
        if (cy0 >= GRID_ROWS) continue;
        const int cy1 = std::min(GRID_ROWS - 1, (int)ceil((y - kfMinY + r) * ghInv));
        if (cy1 < 0) continue;
        int best = 256, bidx = -1;
        for (int ix = cx0; ix <= cx1; ix++)
            for (int iy = cy0; iy <= cy1; iy++) {
                const std::vector<int> &cell = grid[(size_t)ix * GRID_ROWS + iy];
                for (size_t j = 0; j < cell.size(); j++) {
                    const int idx = cell[j];
                    const float *k = kpUn + 7 * (size_t)idx;
                    const float distx = k[0] - x, disty = k[1] - y;
                    if (!(fabs(distx) < r && fabs(disty) < r)) continue;
                    const int kl = (int)k[5];
                    if (kl < lvl - 1 || kl > lvl) continue;
                    if (chi2Gate) {
                        const float ex = x - k[0], ey = y - k[1];
                        if (uRight[idx] >= 0) {
                            const float er = pur[p] - uRight[idx];
                            const float e2 = ex * ex + ey * ey + er * er;
                            if (e2 * invSigma2[kl] > 7.8) continue;
                        } else {
                            const float e2 = ex * ex + ey * ey;
                            if (e2 * invSigma2[kl] > 5.99) continue;
                        }
                    }
                    const int dist = descriptor_distance(pdesc + 32 * (size_t)p, desc + 32 * (size_t)idx);
                    if (dist < best) { best = dist; bidx = idx; }
                }
            }
        bestIdx[p] = bidx; bestDist[p] = best;
    }
}

// ---------------------------------------------------------------------------------------
// The search loops of ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:
// 453-510) and SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:1790-1826) on flat arrays:
// queries in order, GetFeaturesInArea window (window bounds winMinX/Y, cells filed with minX/minY), level gate,
// blocked features skipped, strict '<' minimum, accept when <= maxDist and block the feature.
// ---------------------------------------------------------------------------------------
MO_API int mo_area_search_greedy(const float *kpUn, const uint8_t *desc, const uint8_t *blocked0, int n, float minX, float minY, float winMinX, float winMinY,
                                 float gwInv, float ghInv, const float *qu, const float *qv, const float *qr, const int32_t *qmin, const int32_t *qmax,
                                 const uint8_t *qactive, const uint8_t *qdesc, int m, int maxDist, int32_t *assigned, int32_t *dists)
{
    std::vector<std::vector<int> > grid((size_t)GRID_COLS * GRID_ROWS);
    for (int i = 0; i < n; i++) {
        const int px = (int)round((kpUn[7 * (size_t)i] - minX) * gwInv), py = (int)round((kpUn[7 * (size_t)i + 1] - minY) * ghInv);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
        grid[(size_t)px * GRID_ROWS + py].push_back(i);
    }
    std::vector<char> blocked((size_t)std::max(n, 1), 0);
    for (int i = 0; i < n; i++) blocked[(size_t)i] = blocked0 ? (char)blocked0[i] : 0;
    int nmatches = 0;
    for (int p = 0; p < m; p++) {
        assigned[p] = -1; dists[p] = 256;
        if (qactive && !qactive[p]) continue;
        const float x = qu[p], y = qv[p], r = qr[p];
        const int cx0 = std::max(0, (int)floor((x - winMinX - r) * gwInv));
        if (cx0 >= GRID_COLS) continue;
        const int cx1 = std::min(GRID_COLS - 1, (int)ceil((x - winMinX + r) * gwInv));
        if (cx1 < 0) continue;
        const int cy0 = std::max(0, (int)floor((y - winMinY - r) * ghInv));
        if (cy0 >= GRID_ROWS) continue;
        const int cy1 = std::min(GRID_ROWS - 1, (int)ceil((y - winMinY + r) * ghInv));
        if (cy1 < 0) continue;
        int best = 256, bidx = -1;
        for (int ix = cx0; ix <= cx1; ix++)
            for (int iy = cy0; iy <= cy1; iy++) {
                const std::vector<int> &cell = grid[(size_t)ix * GRID_ROWS + iy];
                for (size_t j = 0; j < cell.size(); j++) {
                    const int idx = cell[j];
                    const float *k = kpUn + 7 * (size_t)idx;
                    const int kl = (int)k[5];
                    if (kl < qmin[p]) continue;
                    if (qmax[p] >= 0 && kl > qmax[p]) continue;
                    const float distx = k[0] - x, disty = k[1] - y;
                    if (!(fabs(distx) < r && fabs(disty) < r)) continue;
                    if (blocked[(size_t)idx]) continue;
                    const int dist = descriptor_distance(qdesc + 32 * (size_t)p, desc + 32 * (size_t)idx);
                    if (dist < best) { best = dist; bidx = idx; }
                }
            }
        if (bidx >= 0 && best <= maxDist) { assigned[p] = bidx; dists[p] = best; blocked[(size_t)bidx] = 1; nmatches++; }   // (maxDist is < 256 in the reference)
    }
    return nmatches;
}

// ---------------------------------------------------------------------------------------
// ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:515-654, on flat arrays (F2 grid filed with
// minX/minY, Frame::GetFeaturesInArea(x, y, windowSize, 0, 0)).  prevXY is vbPrevMatched (not updated here).
// ---------------------------------------------------------------------------------------
MO_API int mo_search_for_initialization(const float *kp1, const uint8_t *desc1, int n1, const float *kp2, const uint8_t *desc2, int n2, float minX, float minY,
                                        float gwInv, float ghInv, const float *prevXY, int windowSize, float nnratio, int checkOri, int32_t *matches12)
{
    std::vector<std::vector<int> > grid((size_t)GRID_COLS * GRID_ROWS);
    for (int i = 0; i < n2; i++) {
        const int px = (int)round((kp2[7 * (size_t)i] - minX) * gwInv), py = (int)round((kp2[7 * (size_t)i + 1] - minY) * ghInv);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
        grid[(size_t)px * GRID_ROWS + py].push_back(i);
    }
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = HISTO_LENGTH / 360.0f;
    std::vector<int> matchedDistance((size_t)std::max(n2, 1), 2147483647), matches21((size_t)std::max(n2, 1), -1);
    for (int i1 = 0; i1 < n1; i1++) {
        const float *k1 = kp1 + 7 * (size_t)i1;
        const int level1 = (int)k1[5];
        if (level1 > 0) continue;
        const float x = prevXY[2 * (size_t)i1], y = prevXY[2 * (size_t)i1 + 1], r = (float)windowSize;
        const int cx0 = std::max(0, (int)floor((x - minX - r) * gwInv));
        if (cx0 >= GRID_COLS) continue;
        const int cx1 = std::min(GRID_COLS - 1, (int)ceil((x - minX + r) * gwInv));
        if (cx1 < 0) continue;
        const int cy0 = std::max(0, (int)floor((y - minY - r) * ghInv));
        if (cy0 >= GRID_ROWS) continue;
        const int cy1 = std::min(GRID_ROWS - 1, (int)ceil((y - minY + r) * ghInv));
        if (cy1 < 0) continue;
        int bestDist = 2147483647, bestDist2 = 2147483647, bestIdx2 = -1;
        for (int ix = cx0; ix <= cx1; ix++)
            for (int iy = cy0; iy <= cy1; iy++) {
                const std::vector<int> &cell = grid[(size_t)ix * GRID_ROWS + iy];
                for (size_t j = 0; j < cell.size(); j++) {
                    const int i2 = cell[j];
                    const float *k2 = kp2 + 7 * (size_t)i2;
                    const int oct = (int)k2[5];
                    if (oct < level1) continue;          // bCheckLevels is true here (maxLevel = 0 >= 0), src/Frame.cc:814-822
                    if (oct > level1) continue;
                    const float distx = k2[0] - x, disty = k2[1] - y;
                    if (!(fabs(distx) < r && fabs(disty) < r)) continue;
                    const int dist = descriptor_distance(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
                    if (matchedDistance[(size_t)i2] <= dist) continue;
                    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
            }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (matches21[(size_t)bestIdx2] >= 0) { matches12[matches21[(size_t)bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                matches21[(size_t)bestIdx2] = i1;
                matchedDistance[(size_t)bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) {
                    float rot = k1[3] - kp2[7 * (size_t)bestIdx2 + 3];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                const int idx1 = rotHist[i][j];
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
            }
        }
    }
    return nmatches;
}

// orbx_bow.hip -- DBoW2 vocabulary-tree descent on gfx950.
//
//   orbx_bow_transform*  ==  DBoW2::TemplatedVocabulary<FORB::TDescriptor,FORB>::transform
//                            (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1262),
//                            the work of Frame::ComputeBoW / KeyFrame::ComputeBoW
//                            (src/Frame.cc:880-896, src/KeyFrame.cc:80-88).
//
// Per feature: from the root pick the child of minimum Hamming distance (FORB::distance,
// FORB.cpp:81-101; first minimum wins, strict '<' at TemplatedVocabulary.h:1219-1229) until a
// leaf: word id + weight, and the node passed at depth L - levelsup = the FeatureVector key that
// gates ORBmatcher::SearchByBoW.  k*L = 60 distances per feature with the stock vocabulary:
// integer VALU + dependent gathers; the tree (35 MB of descriptors at k=10, L=6) lives in HBM /
// Infinity Cache with the children of a node stored contiguously.
// One thread per feature, descriptor in 8 VGPRs.
#include <string.h>

#include <vector>

#include "orbx_internal.h"

namespace {

struct VocNode { int32_t childStart, childCount, word, pad; };   // word >= 0: leaf

__device__ __forceinline__ void bow_descend(const VocNode *__restrict__ nodes, const int32_t *__restrict__ childList, const uint4 *__restrict__ childDesc,
                                            const double *__restrict__ nodeWeight, int nidLevel, const uint8_t *__restrict__ desc, int cap, int32_t *__restrict__ word,
                                            int32_t *__restrict__ node, double *__restrict__ weight, int f, int i, int &wordOut, int &nodeOut)
{
    const uint4 *dp = (const uint4 *)(desc + ((size_t)f * cap + i) * 32);
    const uint4 a0 = dp[0], a1 = dp[1];
    int cur = 0, level = 0, nid = 0;
    VocNode nd = nodes[0];
    while (nd.word < 0) {
        ++level;
        int best = 0x7fffffff, bestPos = nd.childStart;
        for (int c = 0; c < nd.childCount; c++) {
            const uint4 b0 = childDesc[2 * (size_t)(nd.childStart + c)], b1 = childDesc[2 * (size_t)(nd.childStart + c) + 1];
            const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
                          __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
            if (d < best) { best = d; bestPos = nd.childStart + c; }
        }
        cur = childList[bestPos];
        if (level == nidLevel) nid = cur;
        nd = nodes[cur];
    }
    const double w = nodeWeight[cur];
    const size_t o = (size_t)f * cap + i;
    wordOut = nd.word;
    nodeOut = w > 0 ? nid : -1;   // features whose word has weight 0 are not filed in the FeatureVector (:1160-1166)
    word[o] = wordOut;
    weight[o] = w;
    node[o] = nodeOut;
}

// pubFlag != nullptr: the host-array call - word / node / weight are MAPPED host memory and the last workgroup raises the call's sequence word behind them
__global__ __launch_bounds__(256) void k_bow_transform(const VocNode *__restrict__ nodes, const int32_t *__restrict__ childList,
                                                       const uint4 *__restrict__ childDesc, const double *__restrict__ nodeWeight, int nidLevel,
                                                       const uint8_t *__restrict__ desc, const int32_t *__restrict__ counts, int cap, int32_t *__restrict__ word,
                                                       int32_t *__restrict__ node, double *__restrict__ weight, int32_t *__restrict__ wordDev, int32_t *__restrict__ nodeDev,
                                                       unsigned *pubCounter, unsigned long long *pubFlag, unsigned long long pubSeq)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = counts ? min(counts[f], cap) : cap;
    if (i < n) {
        int w = 0, nd = 0;
        bow_descend(nodes, childList, childDesc, nodeWeight, nidLevel, desc, cap, word, node, weight, f, i, w, nd);
        if (wordDev) { wordDev[i] = w; nodeDev[i] = nd; }      // (host-array call, one frame: a device copy for k_bow_ranks - word / node are mapped host memory there)
    }
    if (pubFlag) orbx_publish(pubCounter, pubFlag, pubSeq, gridDim.x * gridDim.y);
}

// The orders of orbx_bow_transform_sorted: rank of feature i among the filed features by (word, i) and by (node, i); 32 features per workgroup, the 8
// lanes of a feature share the scan of the keys (staged in LDS).  Features that are not filed (weight 0: node -1) rank behind all others and are not
// written.  The last workgroup to arrive raises the call's sequence word (the transform's own results were written by the kernel before).
__global__ __launch_bounds__(256) void k_bow_ranks(const int32_t *__restrict__ word, const int32_t *__restrict__ node, int n, int32_t *__restrict__ byWord,
                                                   int32_t *__restrict__ byNode, int32_t *__restrict__ filedOut, unsigned *pubCounter, unsigned long long *pubFlag,
                                                   unsigned long long pubSeq)
{
    __shared__ __attribute__((aligned(16))) int32_t sW[2048], sN[2048];
    const int tid = threadIdx.x, i = blockIdx.x * 32 + (tid >> 3), seg = tid & 7;
    const bool live = i < n;
    const int ni = live ? node[i] : -1, wi = live ? word[i] : 0;
    const bool filedI = live && ni >= 0;
    int rw = 0, rn = 0, filed = 0;
    for (int t0 = 0; t0 < n; t0 += 2048) {
        const int tn = min(2048, n - t0), tn4 = (tn + 3) & ~3;
        __syncthreads();
        // a feature that is not filed (node -1) gets the largest key in both orders: it ranks behind every filed feature and counts for nobody
        for (int k = tid; k < tn4; k += 256) { const int nk = k < tn ? node[t0 + k] : -1; sN[k] = nk >= 0 ? nk : 0x7fffffff; sW[k] = nk >= 0 ? word[t0 + k] : 0x7fffffff; }
        __syncthreads();
#pragma unroll 2
        for (int k = 4 * seg; k < tn4; k += 32) {      // four keys per LDS read (a one-key loop spends its time waiting for LDS: 25 us per call)
            const int4 n4 = *(const int4 *)&sN[k], w4 = *(const int4 *)&sW[k];
            const int kk = t0 + k;
            filed += (n4.x != 0x7fffffff) + (n4.y != 0x7fffffff) + (n4.z != 0x7fffffff) + (n4.w != 0x7fffffff);
            rw += (w4.x < wi || (w4.x == wi && kk < i)) + (w4.y < wi || (w4.y == wi && kk + 1 < i)) + (w4.z < wi || (w4.z == wi && kk + 2 < i)) + (w4.w < wi || (w4.w == wi && kk + 3 < i));
            rn += (n4.x < ni || (n4.x == ni && kk < i)) + (n4.y < ni || (n4.y == ni && kk + 1 < i)) + (n4.z < ni || (n4.z == ni && kk + 2 < i)) + (n4.w < ni || (n4.w == ni && kk + 3 < i));
        }
    }
    rw += __shfl_xor(rw, 1); rw += __shfl_xor(rw, 2); rw += __shfl_xor(rw, 4);
    rn += __shfl_xor(rn, 1); rn += __shfl_xor(rn, 2); rn += __shfl_xor(rn, 4);
    filed += __shfl_xor(filed, 1); filed += __shfl_xor(filed, 2); filed += __shfl_xor(filed, 4);
    if (seg == 0 && filedI) { byWord[rw] = i; byNode[rn] = i; }
    if (blockIdx.x == 0 && tid == 0) *filedOut = filed;
    orbx_publish(pubCounter, pubFlag, pubSeq, gridDim.x);
}

}  // namespace

struct orbx_vocabulary {
    int device = 0, k = 0, L = 0, numNodes = 0, numWords = 0;
    hipStream_t stream = nullptr;   // host-array form only; the device form runs on the extractor's stream
    OrbxDevBuf<VocNode> nodes;
    OrbxDevBuf<int32_t> childList;
    OrbxDevBuf<uint4> childDesc;
    OrbxDevBuf<double> nodeWeight;
    // results, double buffered in lockstep with the extractor's result buffers
    OrbxDevBuf<int32_t> word[2], node[2];
    OrbxDevBuf<double> weight[2];
    int cur = 0, lastBatch = 0, lastCap = 0;
    OrbxCallBox box;   // host-array form: descriptors in, word / node / weight out through mapped pinned memory
};

extern "C" int orbx_vocabulary_create(int device, int k, int L, int num_nodes, const int32_t *parent, const uint8_t *is_leaf, const uint8_t *descriptors,
                                      const double *weights, orbx_vocabulary **out)
{
    if (!out || !parent || !is_leaf || !descriptors || !weights || num_nodes < 2 || k < 1 || L < 1) { orbx_set_error("bad vocabulary arguments"); return ORBX_ERR_ARG; }
    *out = nullptr;
    std::vector<std::vector<int32_t> > children((size_t)num_nodes);
    for (int i = 1; i < num_nodes; i++) {
        if (parent[i] < 0 || parent[i] >= i) { orbx_set_error("node %d: parent %d must be a smaller node id", i, parent[i]); return ORBX_ERR_ARG; }
        if (is_leaf[parent[i]]) { orbx_set_error("node %d hangs under leaf %d", i, parent[i]); return ORBX_ERR_ARG; }
        children[(size_t)parent[i]].push_back(i);   // ascending id = the order loadFromTextFile builds (TemplatedVocabulary.h:1378-1420)
    }
    std::vector<VocNode> nodes((size_t)num_nodes);
    std::vector<int32_t> childList;
    int words = 0;
    for (int i = 0; i < num_nodes; i++) {
        nodes[(size_t)i].childStart = (int32_t)childList.size();
        nodes[(size_t)i].childCount = (int32_t)children[(size_t)i].size();
        nodes[(size_t)i].pad = 0;
        const bool leaf = i > 0 && is_leaf[i];
        if (!leaf && children[(size_t)i].empty()) { orbx_set_error("inner node %d has no children", i); return ORBX_ERR_ARG; }
        nodes[(size_t)i].word = leaf ? words++ : -1;   // word ids in node-id order (:1407-1414)
        childList.insert(childList.end(), children[(size_t)i].begin(), children[(size_t)i].end());
    }
    std::vector<uint8_t> cdesc(childList.size() * 32);
    for (size_t c = 0; c < childList.size(); c++) memcpy(&cdesc[32 * c], descriptors + 32 * (size_t)childList[c], 32);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { orbx_set_error("no HIP device available: liborbx has no CPU fallback"); return ORBX_ERR_NODEVICE; }
    if (device < 0 || device >= ndev) { orbx_set_error("device %d out of range", device); return ORBX_ERR_ARG; }
    ORBX_HIP_CHECK(hipSetDevice(device));
    orbx_vocabulary *v = new orbx_vocabulary();
    v->device = device; v->k = k; v->L = L; v->numNodes = num_nodes; v->numWords = words;
    int rc;
    if (hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) != hipSuccess) { delete v; orbx_set_error("hipStreamCreate failed"); return ORBX_ERR_HIP; }
    if ((rc = v->nodes.ensure(nodes.size())) || (rc = v->childList.ensure(childList.size())) || (rc = v->childDesc.ensure(childList.size() * 2)) ||
        (rc = v->nodeWeight.ensure((size_t)num_nodes))) {
        orbx_vocabulary_destroy(v);
        return rc;
    }
    ORBX_HIP_CHECK(hipMemcpy(v->nodes.p, nodes.data(), nodes.size() * sizeof(VocNode), hipMemcpyHostToDevice));
    ORBX_HIP_CHECK(hipMemcpy(v->childList.p, childList.data(), childList.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    ORBX_HIP_CHECK(hipMemcpy(v->childDesc.p, cdesc.data(), cdesc.size(), hipMemcpyHostToDevice));
    ORBX_HIP_CHECK(hipMemcpy(v->nodeWeight.p, weights, (size_t)num_nodes * sizeof(double), hipMemcpyHostToDevice));
    *out = v;
    return ORBX_OK;
}

extern "C" void orbx_vocabulary_destroy(orbx_vocabulary *v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->stream) { (void)hipStreamSynchronize(v->stream); (void)hipStreamDestroy(v->stream); }
    v->nodes.release(); v->childList.release(); v->childDesc.release(); v->nodeWeight.release(); v->box.release();
    for (int b = 0; b < 2; b++) { v->word[b].release(); v->node[b].release(); v->weight[b].release(); }
    delete v;
}

extern "C" int orbx_vocabulary_words(const orbx_vocabulary *v) { return v ? v->numWords : ORBX_ERR_ARG; }

static int launch_transform(orbx_vocabulary *v, hipStream_t stream, const uint8_t *desc, const int32_t *counts, int batch, int cap, int levelsup)
{
    int rc;
    v->cur ^= 1;
    const int b = v->cur;
    const size_t n = (size_t)batch * cap;
    if ((rc = v->word[b].ensure(n)) || (rc = v->node[b].ensure(n)) || (rc = v->weight[b].ensure(n))) return rc;
    hipLaunchKernelGGL(k_bow_transform, dim3((unsigned)((cap + 255) / 256), (unsigned)batch), dim3(256), 0, stream, v->nodes.p, v->childList.p, v->childDesc.p,
                       v->nodeWeight.p, v->L - levelsup, desc, counts, cap, v->word[b].p, v->node[b].p, v->weight[b].p, (int32_t *)nullptr, (int32_t *)nullptr, (unsigned *)nullptr, (unsigned long long *)nullptr, 0ull);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { orbx_set_error("kernel launch failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
    v->lastBatch = batch; v->lastCap = cap;
    return ORBX_OK;
}

extern "C" int orbx_bow_transform_device(orbx_vocabulary *v, orbx_extractor *ext, int levelsup)
{
    if (!v || !ext) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    OrbxLastBatchView view;
    int rc = orbx_extractor_last_batch_view_internal(ext, &view);
    if (rc != ORBX_OK) return rc;
    ORBX_HIP_CHECK(hipSetDevice(v->device));
    // same stream as the extraction: ordered behind it, and ahead of every consumer that orders itself behind the extractor
    return launch_transform(v, orbx_extractor_stream_internal(ext), view.desc, view.counts, view.batch, view.cap, levelsup);
}

extern "C" int orbx_bow_results_device(orbx_vocabulary *v, const int32_t **word_dev, const int32_t **node_dev, const double **weight_dev, int *capacity)
{
    if (!v) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (!v->lastBatch) { orbx_set_error("no transform has run yet"); return ORBX_ERR_STATE; }
    if (word_dev) *word_dev = v->word[v->cur].p;
    if (node_dev) *node_dev = v->node[v->cur].p;
    if (weight_dev) *weight_dev = v->weight[v->cur].p;
    if (capacity) *capacity = v->lastCap;
    return ORBX_OK;
}

extern "C" int orbx_bow_download(orbx_vocabulary *v, orbx_extractor *ext, int batch, int32_t *word, int32_t *node, double *weight)
{
    if (!v) { orbx_set_error("NULL handle"); return ORBX_ERR_ARG; }
    if (batch < 1 || batch > v->lastBatch) { orbx_set_error("batch not available"); return ORBX_ERR_STATE; }
    ORBX_HIP_CHECK(hipSetDevice(v->device));
    ORBX_HIP_CHECK(hipStreamSynchronize(ext ? orbx_extractor_stream_internal(ext) : v->stream));
    const size_t n = (size_t)batch * v->lastCap;
    const int b = v->cur;
    if (word) ORBX_HIP_CHECK(hipMemcpy(word, v->word[b].p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (node) ORBX_HIP_CHECK(hipMemcpy(node, v->node[b].p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (weight) ORBX_HIP_CHECK(hipMemcpy(weight, v->weight[b].p, n * sizeof(double), hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// Host-array form (Frame::ComputeBoW / KeyFrame::ComputeBoW through shim/BoW_hip.cc): the descent kernel reads the descriptors from mapped pinned memory
// (32 bytes per thread, each once) and writes word / node / weight into mapped pinned memory (and word / node into device memory for the ranking
// kernel of the _sorted form); no copy engine, no stream synchronisation (OrbxCallBox).
static int bow_transform_host(orbx_vocabulary *v, const uint8_t *descriptors, int n, int levelsup, int32_t *word, int32_t *node, double *weight, int32_t *by_word, int32_t *by_node,
                              int32_t *filed)
{
    if (!v || (n > 0 && !descriptors)) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    if (filed) *filed = 0;
    if (n <= 0) return ORBX_OK;
    ORBX_HIP_CHECK(hipSetDevice(v->device));
    OrbxCallBox &bx = v->box;
    const bool sorted = by_word && by_node && filed;
    const size_t N = (size_t)n, offNode = bx.padded(N * 4), offW = offNode + bx.padded(N * 4), offBW = offW + bx.padded(N * 8), offBN = offBW + bx.padded(N * 4),
                 offF = offBN + bx.padded(N * 4);
    int rc = bx.begin(bx.padded(N * 32), offF + bx.padded(4), v->stream);
    if (rc != ORBX_OK) return rc;
    const uint8_t *dDesc = bx.put(descriptors, N * 32);
    const unsigned long long seq = bx.arm();
    if (sorted && ((rc = v->word[0].ensure(N)) || (rc = v->node[0].ensure(N)))) return rc;
    hipLaunchKernelGGL(k_bow_transform, dim3((unsigned)((n + 255) / 256), 1u), dim3(256), 0, v->stream, v->nodes.p, v->childList.p, v->childDesc.p, v->nodeWeight.p, v->L - levelsup,
                       dDesc, (const int32_t *)nullptr, n, bx.outDev<int32_t>(0), bx.outDev<int32_t>(offNode), bx.outDev<double>(offW), sorted ? v->word[0].p : (int32_t *)nullptr,
                       sorted ? v->node[0].p : (int32_t *)nullptr, bx.counter, sorted ? (unsigned long long *)nullptr : bx.flagDev, seq);
    if (sorted)
        hipLaunchKernelGGL(k_bow_ranks, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, v->stream, (const int32_t *)v->word[0].p, (const int32_t *)v->node[0].p, n, bx.outDev<int32_t>(offBW),
                           bx.outDev<int32_t>(offBN), bx.outDev<int32_t>(offF), bx.counter, bx.flagDev, seq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { orbx_set_error("kernel launch failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
    if ((rc = bx.wait(v->stream)) != ORBX_OK) return rc;
    if (word) memcpy(word, bx.outHost<int32_t>(0), N * 4);
    if (node) memcpy(node, bx.outHost<int32_t>(offNode), N * 4);
    if (weight) memcpy(weight, bx.outHost<double>(offW), N * 8);
    if (sorted) {
        const int nf = *bx.outHost<int32_t>(offF);
        *filed = nf;
        memcpy(by_word, bx.outHost<int32_t>(offBW), (size_t)nf * 4);
        memcpy(by_node, bx.outHost<int32_t>(offBN), (size_t)nf * 4);
    }
    return ORBX_OK;
}

extern "C" int orbx_bow_transform(orbx_vocabulary *v, const uint8_t *descriptors, int n, int levelsup, int32_t *word, int32_t *node, double *weight)
{
    return bow_transform_host(v, descriptors, n, levelsup, word, node, weight, nullptr, nullptr, nullptr);
}

extern "C" int orbx_bow_transform_sorted(orbx_vocabulary *v, const uint8_t *descriptors, int n, int levelsup, int32_t *word, int32_t *node, double *weight, int32_t *by_word,
                                         int32_t *by_node, int32_t *filed)
{
    if (!by_word || !by_node || !filed) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    return bow_transform_host(v, descriptors, n, levelsup, word, node, weight, by_word, by_node, filed);
}

// ---- the latency form (include/orbx.h: orbx_bow_job_*) ----
struct orbx_bow_job {
    orbx_vocabulary *v = nullptr;
    int device = 0;                 // (its own copy: a job may be destroyed after its vocabulary - the shim does, when a vocabulary object was replaced at the same address)
    hipStream_t stream = nullptr;
    OrbxCallBox box;
    OrbxDevBuf<int32_t> word, node;
    int pendingN = -1;      // -1: nothing begun
    size_t offNode = 0, offW = 0, offBW = 0, offBN = 0, offF = 0;
};

extern "C" int orbx_bow_job_create(orbx_vocabulary *v, orbx_bow_job **out)
{
    if (!v || !out) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    *out = nullptr;
    ORBX_HIP_CHECK(hipSetDevice(v->device));
    orbx_bow_job *j = new orbx_bow_job();
    j->v = v; j->device = v->device;
    if (hipStreamCreateWithFlags(&j->stream, hipStreamNonBlocking) != hipSuccess) { delete j; orbx_set_error("hipStreamCreate failed"); return ORBX_ERR_HIP; }
    *out = j;
    return ORBX_OK;
}

extern "C" void orbx_bow_job_destroy(orbx_bow_job *j)
{
    if (!j) return;
    (void)hipSetDevice(j->device);
    if (j->stream) { (void)hipStreamSynchronize(j->stream); (void)hipStreamDestroy(j->stream); }
    j->box.release(); j->word.release(); j->node.release();
    delete j;
}

extern "C" int orbx_bow_job_begin(orbx_bow_job *j, orbx_extractor *ext, int levelsup)
{
    if (!j || !ext) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    j->pendingN = -1;
    int st = 0;
    if (!orbx_extractor_host_complete_internal(ext, &st)) { orbx_set_error("the extractor's last call was not a completed single-frame call"); return ORBX_ERR_STATE; }
    if (st) { orbx_set_error("the extractor call these features come from overflowed a device capacity (bits 0x%x): results are not the reference's", st); return ORBX_ERR_CAPACITY; }
    OrbxLastBatchView view;
    int rc = orbx_extractor_last_batch_view_internal(ext, &view);
    if (rc != ORBX_OK) return rc;
    orbx_vocabulary *v = j->v;
    ORBX_HIP_CHECK(hipSetDevice(v->device));
    const int n = orbx_extractor_host_count_internal(ext), cap = view.cap;
    if (n < 0 || n > cap) { orbx_set_error("feature count %d out of range", n); return ORBX_ERR_CAPACITY; }
    if (n == 0) { j->pendingN = 0; return ORBX_OK; }
    OrbxCallBox &bx = j->box;
    const size_t N = (size_t)n;
    j->offNode = bx.padded(N * 4); j->offW = j->offNode + bx.padded(N * 4); j->offBW = j->offW + bx.padded(N * 8); j->offBN = j->offBW + bx.padded(N * 4); j->offF = j->offBN + bx.padded(N * 4);
    if ((rc = bx.begin(0, j->offF + bx.padded(4), j->stream)) != ORBX_OK) return rc;
    if ((rc = j->word.ensure(N)) || (rc = j->node.ensure(N))) return rc;
    const unsigned long long seq = bx.arm();
    // (frame 0 of the view, n features: `cap` = n for the kernel's indexing - one frame, feature i at i)
    hipLaunchKernelGGL(k_bow_transform, dim3((unsigned)((n + 255) / 256), 1u), dim3(256), 0, j->stream, v->nodes.p, v->childList.p, v->childDesc.p, v->nodeWeight.p, v->L - levelsup,
                       view.desc, (const int32_t *)nullptr, n, bx.outDev<int32_t>(0), bx.outDev<int32_t>(j->offNode), bx.outDev<double>(j->offW), j->word.p, j->node.p, bx.counter,
                       (unsigned long long *)nullptr, seq);
    hipLaunchKernelGGL(k_bow_ranks, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, j->stream, (const int32_t *)j->word.p, (const int32_t *)j->node.p, n, bx.outDev<int32_t>(j->offBW),
                       bx.outDev<int32_t>(j->offBN), bx.outDev<int32_t>(j->offF), bx.counter, bx.flagDev, seq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { orbx_set_error("kernel launch failed: %s", hipGetErrorString(e)); return ORBX_ERR_HIP; }
    j->pendingN = n;
    return ORBX_OK;
}

extern "C" int orbx_bow_job_end(orbx_bow_job *j, const int32_t **word, const int32_t **node, const double **weight, const int32_t **by_word, const int32_t **by_node,
                                int32_t *filed, int32_t *n)
{
    if (!j || !filed || !n) { orbx_set_error("NULL argument"); return ORBX_ERR_ARG; }
    const int pn = j->pendingN;
    j->pendingN = -1;
    if (pn < 0) { orbx_set_error("no job has been begun"); return ORBX_ERR_STATE; }
    *n = pn; *filed = 0;
    if (pn == 0) return ORBX_OK;
    ORBX_HIP_CHECK(hipSetDevice(j->device));
    int rc = j->box.wait(j->stream);
    if (rc != ORBX_OK) return rc;
    const OrbxCallBox &bx = j->box;
    if (word) *word = bx.outHost<int32_t>(0);
    if (node) *node = bx.outHost<int32_t>(j->offNode);
    if (weight) *weight = bx.outHost<double>(j->offW);
    if (by_word) *by_word = bx.outHost<int32_t>(j->offBW);
    if (by_node) *by_node = bx.outHost<int32_t>(j->offBN);
    *filed = *bx.outHost<int32_t>(j->offF);
    return ORBX_OK;
}

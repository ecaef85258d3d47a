// shim/BoW_hip.cc -- HIP bodies for ORB_SLAM2::Frame::ComputeBoW and KeyFrame::ComputeBoW.
//
// Compiled against the REFERENCE's own headers.  Replaces the bodies of
//     void Frame::ComputeBoW()        src/Frame.cc:880-896
//     void KeyFrame::ComputeBoW()     src/KeyFrame.cc:80-88
// i.e. `mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)`: the k*L Hamming
// distances per feature of the tree descent run on the MI355X (orbx_bow_transform, csrc/orbx_bow.hip),
// the two std::map results are then filled with DBoW2's own BowVector / FeatureVector methods in
// the order of the reference's loop (TemplatedVocabulary.h:1146-1196), so they are bit-identical.
// The device copy of the tree is built once per ORBVocabulary object from its (protected) node
// table, reached through a pointer-to-member of a derived accessor - no change to DBoW2.
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
#include "orbx.h"
#include "shim_error.h"

static unsigned long gBoWCalls = 0, gEarlyBoW = 0;
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_compute_bow_calls(void) { return gBoWCalls; }
extern "C" __attribute__((visibility("default"))) unsigned long orbx_shim_early_bow(void) { return gEarlyBoW; }      // ComputeBoW calls served by a job the constructor began
// shim/Frame_hip.cc (when it is part of the build): the job it began for frame `frameId` of `leftExtractor` with nFeatures features, or NULL; one shot
extern "C" __attribute__((weak)) void *orbx_shim_early_bow_take(void *leftExtractor, long frameId, int nFeatures);

namespace ORB_SLAM2
{

namespace
{
struct VocAccess : public ORBVocabulary {
    static DBoW2::GeneralScoring *Scoring(const ORBVocabulary &v) { return v.*(&VocAccess::m_scoring_object); }
    // the node table as the flat arrays of orbx_vocabulary_create (Node is a protected nested type)
    // Cheap identity of the tree's CONTENT: size, shape and 64 sampled nodes (parent, weight, descriptor).  The device copy is cached per
    // ORBVocabulary address; the reference keeps one vocabulary for the life of the process, but an object destroyed and another one
    // created at the same address (tests do that) must not be served the old tree.
    static unsigned long long Fingerprint(const ORBVocabulary &v)
    {
        const std::vector<Node> &nodes = v.*(&VocAccess::m_nodes);
        unsigned long long h = 1469598103934665603ull;
        const auto mix = [&h](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
        const size_t n = nodes.size();
        const int k = v.getBranchingFactor(), L = v.getDepthLevels();
        mix(&n, sizeof(n)); mix(&k, sizeof(k)); mix(&L, sizeof(L));
        for (size_t s = 0; s < 64 && n > 0; s++) {
            const Node &nd = nodes[(size_t)((unsigned long long)s * (n - 1) / 63)];
            mix(&nd.parent, sizeof(nd.parent)); mix(&nd.weight, sizeof(nd.weight));
            if (!nd.descriptor.empty()) mix(nd.descriptor.ptr<unsigned char>(), 32);
        }
        return h;
    }
    static void Flatten(const ORBVocabulary &v, std::vector<int32_t> &parent, std::vector<uint8_t> &leaf, std::vector<uint8_t> &desc, std::vector<double> &weight)
    {
        const std::vector<Node> &nodes = v.*(&VocAccess::m_nodes);
        const size_t n = nodes.size();
        parent.resize(n); leaf.resize(n); weight.resize(n); desc.assign(n * 32, 0);
        for (size_t i = 0; i < n; i++) {
            parent[i] = (int32_t)nodes[i].parent;
            leaf[i] = (i > 0 && nodes[i].isLeaf()) ? 1 : 0;
            weight[i] = nodes[i].weight;
            if (i > 0 && !nodes[i].descriptor.empty()) memcpy(&desc[32 * i], nodes[i].descriptor.ptr<unsigned char>(), 32);
        }
    }
};

// One device vocabulary per ORBVocabulary object.  An orbx_vocabulary handle is NOT re-entrant (its scratch / result buffers are
// member state, like every orbx handle: "one call at a time per handle", include/orbx.h), but Tracking (Frame::ComputeBoW) and
// LocalMapping / LoopClosing (KeyFrame::ComputeBoW) share the vocabulary: `call` serialises the orbx_bow_transform calls on it.
struct DeviceVoc { orbx_vocabulary *h; unsigned long long fp, generation; std::mutex call; DeviceVoc() : h(0), fp(0), generation(0) {} };
std::mutex gVocMutex;
std::map<const ORBVocabulary *, DeviceVoc *> gVocs;

DeviceVoc *DeviceVocabulary(const ORBVocabulary *voc)
{
    std::unique_lock<std::mutex> lock(gVocMutex);
    const unsigned long long fp = VocAccess::Fingerprint(*voc);
    std::map<const ORBVocabulary *, DeviceVoc *>::iterator it = gVocs.find(voc);
    if (it != gVocs.end()) {
        if (it->second->fp == fp) return it->second;
        // another vocabulary now lives at this address: rebuild the device copy (wait for a call still running on the old one)
        std::unique_lock<std::mutex> call(it->second->call);
        if (it->second->h) { orbx_vocabulary_destroy(it->second->h); it->second->h = 0; }
    }
    std::vector<int32_t> parent;
    std::vector<uint8_t> leaf, desc;
    std::vector<double> weight;
    VocAccess::Flatten(*voc, parent, leaf, desc, weight);
    const int n = (int)parent.size();
    orbx_vocabulary *dv = 0;
    if (orbx_vocabulary_create(orbx_shim::Device(), voc->getBranchingFactor(), voc->getDepthLevels(), n, &parent[0], &leaf[0], &desc[0], &weight[0], &dv) != ORBX_OK) {
        orbx_shim::Fail("ComputeBoW");
        return 0;      // (not cached: the next call tries again)
    }
    DeviceVoc *d = it != gVocs.end() ? it->second : new DeviceVoc();
    d->h = dv; d->fp = fp; d->generation++;
    gVocs[voc] = d;
    return d;
}

// The reference's loop over the features (TemplatedVocabulary.h:1146-1196) from the device's results: word / node / weight per feature and the two key orders.
void FillMaps(const ORBVocabulary *voc, const int32_t *word, const int32_t *node, const double *weight, const int32_t *byWord, const int32_t *byNode, int filed,
              DBoW2::BowVector &v, DBoW2::FeatureVector &fv)
{
    DBoW2::LNorm norm;
    const bool must = VocAccess::Scoring(*voc)->mustNormalize(norm);
    const DBoW2::WeightingType wt = voc->getWeightingType();
    const bool tf = wt == DBoW2::TF || wt == DBoW2::TF_IDF;
    // The reference's loop over the features (:1146-1196) does v.addWeight / v.addIfNotExist (word) and fv.addFeature (node, i) for every feature with a
    // positive weight, in feature order: a tree search per call.  The same maps from the device's two orders: keys arrive ascending, so every new key goes
    // in at the end of the map; the weights of a word are added in feature order (addWeight's sums, bit for bit), addIfNotExist keeps the first, and a
    // node's feature list is appended in feature order.
    {
        DBoW2::BowVector::iterator vit = v.end();
        for (int k = 0; k < filed; k++) {
            const int i = byWord[k];
            const DBoW2::WordId w = (DBoW2::WordId)word[i];
            if (vit == v.end() || vit->first != w) vit = v.insert(v.end(), DBoW2::BowVector::value_type(w, weight[i]));      // :1160 / :1187 (new word)
            else if (tf) vit->second += weight[i];                                                                          // addWeight on an existing word
        }
        // a node's features are one run of byNode: its vector is made at its final size (appending one feature at a time reallocates it four or five times
        // per node - ORBvoc at levelsup 4 files a frame's features under ~100 nodes)
        for (int k = 0; k < filed;) {
            const DBoW2::NodeId nd = (DBoW2::NodeId)node[byNode[k]];
            int e = k + 1;
            while (e < filed && (DBoW2::NodeId)node[byNode[e]] == nd) e++;
            DBoW2::FeatureVector::iterator fit = fv.insert(fv.end(), DBoW2::FeatureVector::value_type(nd, std::vector<unsigned int>()));
            std::vector<unsigned int> &lst = fit->second;
            lst.resize((size_t)(e - k));
            for (int q = k; q < e; q++) lst[(size_t)(q - k)] = (unsigned int)byNode[q];                                                 // :1161, in feature order
            k = e;
        }
    }
    if (tf && !v.empty() && !must) {                                                          // :1165-1171
        const double nd = v.size();
        for (DBoW2::BowVector::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
    }
    if (must) v.normalize(norm);                                                              // :1196
}

// TemplatedVocabulary::transform(features, v, fv, levelsup), :1127-1196, with the per-feature
// descent done on the device.
void Transform(const ORBVocabulary *voc, const cv::Mat &descriptors, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup)
{
    __atomic_add_fetch(&gBoWCalls, 1, __ATOMIC_RELAXED);
    orbx_shim::Mark("ComputeBoW enters");
    v.clear();
    fv.clear();
    if (voc->empty()) return;
    const int n = descriptors.rows;
    std::vector<int32_t> word((size_t)(n > 0 ? n : 1)), node((size_t)(n > 0 ? n : 1)), byWord((size_t)(n > 0 ? n : 1)), byNode((size_t)(n > 0 ? n : 1));
    std::vector<double> weight((size_t)(n > 0 ? n : 1));
    int32_t filed = 0;
    // (the library copies the descriptors into its mapped pinned buffer itself: a continuous matrix - what the extractor and KeyFrame's clone produce - goes as it is)
    std::vector<unsigned char> flat;
    const unsigned char *rows = descriptors.data;
    if (n > 0 && !(descriptors.isContinuous() && descriptors.cols == 32)) {
        flat.resize((size_t)n * 32);
        for (int i = 0; i < n; i++) memcpy(&flat[32 * (size_t)i], descriptors.ptr<unsigned char>(i), 32);
        rows = &flat[0];
    }
    {
        DeviceVoc *dv = DeviceVocabulary(voc);
        if (!dv) return;      // empty vectors, as for an image without features
        std::unique_lock<std::mutex> call(dv->call);      // the whole call: descent, ranking, results into OUR vectors
        if (orbx_bow_transform_sorted(dv->h, rows, n, levelsup, &word[0], &node[0], &weight[0], &byWord[0], &byNode[0], &filed) != ORBX_OK) { orbx_shim::Fail("ComputeBoW"); return; }
    }
    orbx_shim::Mark("ComputeBoW device call returned");
    FillMaps(voc, &word[0], &node[0], &weight[0], &byWord[0], &byNode[0], filed, v, fv);
    orbx_shim::Mark("ComputeBoW returns");
}
}  // namespace

}  // namespace ORB_SLAM2

// for shim/Frame_hip.cc: the device copy of `voc` (an ORBVocabulary *) and the number of times it has been (re)built for that address
extern "C" __attribute__((visibility("default"))) orbx_vocabulary *orbx_shim_device_vocabulary(const void *voc, unsigned long long *generation)
{
    const ORB_SLAM2::ORBVocabulary *v = (const ORB_SLAM2::ORBVocabulary *)voc;
    if (!v || v->empty()) return 0;
    ORB_SLAM2::DeviceVoc *d = ORB_SLAM2::DeviceVocabulary(v);
    if (!d) return 0;
    if (generation) *generation = d->generation;
    return d->h;
}

namespace ORB_SLAM2
{

void Frame::ComputeBoW()
{
    if (!mBowVec.empty()) return;      // src/Frame.cc:883
    // the descent of THIS frame's descriptors may have been started from inside its constructor, on the device-resident features the moment the
    // extraction was complete (shim/Frame_hip.cc: PostExtract -> orbx_bow_job_begin): then only the maps are left to fill
    if (orbx_shim_early_bow_take && mpORBvocabulary && !mpORBvocabulary->empty()) {
        orbx_bow_job *job = (orbx_bow_job *)orbx_shim_early_bow_take(mpORBextractorLeft, (long)mnId, N);
        if (job) {
            orbx_shim::Mark("ComputeBoW enters (begun by the constructor)");
            const int32_t *word = 0, *node = 0, *byWord = 0, *byNode = 0;
            const double *weight = 0;
            int32_t filed = 0, n = 0;
            if (orbx_bow_job_end(job, &word, &node, &weight, &byWord, &byNode, &filed, &n) == ORBX_OK && n == N) {
                __atomic_add_fetch(&gBoWCalls, 1, __ATOMIC_RELAXED);
                __atomic_add_fetch(&gEarlyBoW, 1, __ATOMIC_RELAXED);
                orbx_shim::Mark("ComputeBoW device call returned");
                mBowVec.clear(); mFeatVec.clear();
                if (n > 0) FillMaps(mpORBvocabulary, word, node, weight, byWord, byNode, filed, mBowVec, mFeatVec);
                orbx_shim::Mark("ComputeBoW returns");
                return;
            }
        }
    }
    Transform(mpORBvocabulary, mDescriptors, mBowVec, mFeatVec, 4);      // :883-894
}

void KeyFrame::ComputeBoW()
{
    if (mBowVec.empty() || mFeatVec.empty()) Transform(mpORBvocabulary, mDescriptors, mBowVec, mFeatVec, 4);   // src/KeyFrame.cc:82-87
}

}  // namespace ORB_SLAM2

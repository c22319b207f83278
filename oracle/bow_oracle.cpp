// TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
//
// CPU restatement of the bag-of-words transform that data::frame::compute_bow / data::keyframe::compute_bow call
// (src/PLPSLAM/data/frame.cc:785-795: bow_vocab_->transform(to_desc_vec(descriptors_), bow_vec_, bow_feat_vec_, 4);
// SURVEY.md 8(f) item 3).  The algorithm lives in DBoW2 (OpenVSLAM's fork, built with USE_DBOW2,
// src/PLPSLAM/CMakeLists.txt:115), a dependency that is NOT under /root/reference: restated from the published
// TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features, BowVector&, FeatureVector&, levelsup) and the
// single-feature transform(feature, word_id, weight, nid, levelsup) = PARITY UNPINNED.  The reference's own consumers pin
// the shape of the result: bow_tree.cc:52-150 walks the two std::map<NodeId, std::vector<unsigned>> in key order.
//   * descent: from the root, the child with the smallest Hamming distance, the FIRST one in the parent's child order on
//     ties (`d < best_d`); the node passed at level L - levelsup is the feature's NodeId (root when L - levelsup <= 0);
//   * BowVector: std::map<WordId, double>; TF / TF_IDF add the word's weight once per feature in feature order, IDF /
//     BINARY keep one copy; a word of weight <= 0 ("stopped") contributes neither a weight nor a feature;
//   * without a normalising scoring (DOT_PRODUCT) TF / TF_IDF divide by the number of distinct words; L1 / L2 norms are
//     accumulated in word order and divide every entry when > 0.
// A leaf above level L - levelsup leaves `nid` unset in DBoW2 (an uninitialised local); here it gets the leaf's node id.
// The tree is passed flat: children of node i are children[child_offset[i] .. child_offset[i + 1]) in their stored order.
#include <cmath>
#include <cstdint>
#include <map>
#include <vector>

namespace {

int hamming32(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}

}  // namespace

extern "C" {

// accumulate: 1 for TF / TF_IDF, 0 for IDF / BINARY.  norm: 0 none, 1 L1, 2 L2.
// Per-feature outputs: word_id / node_id (0xFFFFFFFF for a stopped word).  BowVector: bow_word / bow_value (size *n_bow).
// FeatureVector flattened in map order: fv_node / fv_feat (size *n_fv), features of one node in increasing index.
void oracle_bow_transform(int n_nodes, int L, const int32_t* child_offset, const int32_t* children, const uint8_t* node_desc, const double* node_weight,
                          const uint32_t* node_word, const uint8_t* desc, int n, int levelsup, int accumulate, int norm, uint32_t* word_id,
                          uint32_t* node_id, uint32_t* bow_word, double* bow_value, int* n_bow, uint32_t* fv_node, uint32_t* fv_feat, int* n_fv) {
    std::map<uint32_t, double> v;
    std::map<uint32_t, std::vector<uint32_t>> fv;
    *n_bow = 0; *n_fv = 0;
    if (n_nodes <= 1) return;    // empty vocabulary
    const int nid_level = L - levelsup;
    for (int i = 0; i < n; ++i) {
        const uint8_t* f = desc + (size_t)i * 32;
        uint32_t final_id = 0, nid = 0;
        bool nid_set = nid_level <= 0;
        int current_level = 0;
        do {
            ++current_level;
            const int32_t* ch = children + child_offset[final_id];
            const int nc = child_offset[final_id + 1] - child_offset[final_id];
            final_id = (uint32_t)ch[0];
            double best_d = hamming32(f, node_desc + (size_t)final_id * 32);
            for (int c = 1; c < nc; ++c) {
                const uint32_t id = (uint32_t)ch[c];
                const double d = hamming32(f, node_desc + (size_t)id * 32);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) { nid = final_id; nid_set = true; }
        } while (child_offset[final_id + 1] != child_offset[final_id]);
        if (!nid_set) nid = final_id;
        const double w = node_weight[final_id];
        if (w > 0) {
            const uint32_t id = node_word[final_id];
            auto it = v.lower_bound(id);
            if (it != v.end() && !(v.key_comp()(id, it->first))) {
                if (accumulate) it->second += w;
            } else {
                v.insert(it, {id, w});
            }
            fv[nid].push_back((uint32_t)i);
            word_id[i] = id; node_id[i] = nid;
        } else {
            word_id[i] = 0xFFFFFFFFu; node_id[i] = 0xFFFFFFFFu;
        }
    }
    if (accumulate && !v.empty() && norm == 0) {
        const double nd = (double)v.size();
        for (auto& e : v) e.second /= nd;
    }
    if (norm != 0) {
        double s = 0.0;
        if (norm == 1) { for (auto& e : v) s += std::fabs(e.second); }
        else { for (auto& e : v) s += e.second * e.second; s = std::sqrt(s); }
        if (s > 0.0) for (auto& e : v) e.second /= s;
    }
    int k = 0;
    for (auto& e : v) { bow_word[k] = e.first; bow_value[k] = e.second; ++k; }
    *n_bow = k;
    k = 0;
    for (auto& e : fv) for (uint32_t fi : e.second) { fv_node[k] = e.first; fv_feat[k] = fi; ++k; }
    *n_fv = k;
}

}  // extern "C"

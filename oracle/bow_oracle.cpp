// ORACLE — test infrastructure only (see orb_oracle.cpp header). CPU restatement of Frame::ComputeBoW (src/Frame.cc:1053-1060)
// = DBoW2 TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup)
//   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1195 (TF_IDF weighting) and :1218-1260 (tree descent),
//   BowVector::addWeight / normalize(L1)   Thirdparty/DBoW2/DBoW2/BowVector.cpp:34-46, 62-84,
//   FeatureVector::addFeature               Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:31-45,
//   FORB::distance                          Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101.
// The vocabulary (ORBvoc.txt, weighting TF_IDF, scoring L1_NORM) is a host object; it is passed here as flat arrays: for
// node id: child_begin/child_count into child_ids (children in the order of Node::children), a 32-byte descriptor, the
// weight, the word id; a node without children is a leaf.  L = depth of the leaves.
// PARITY UNPINNED: ORBvoc.txt is absent offline; the tests use synthetic vocabularies.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <vector>

#include "psl_oracle.h"

extern "C" {

// Outputs: per feature word/weight/nid (weight <= 0: the feature is dropped, as the reference does for stopped words);
// BowVector as ascending (word id, value) pairs; FeatureVector as ascending node ids with start offsets into fv_idx.
// Returns the number of BowVector entries; *n_fv = number of FeatureVector nodes.
int pso_compute_bow(const int32_t* child_begin, const int32_t* child_count, const int32_t* child_ids, const uint8_t* node_desc,
                    const double* node_weight, const int32_t* node_word, int L, int levelsup, const uint8_t* desc, int n, int32_t* f_word,
                    double* f_weight, int32_t* f_nid, int32_t* bow_id, double* bow_val, int32_t* fv_node, int32_t* fv_start, int32_t* fv_idx,
                    int* n_fv) {
    std::map<int32_t, double> v;
    std::map<int32_t, std::vector<int32_t>> fv;
    const int nid_level = L - levelsup;
    for (int i = 0; i < n; ++i) {
        const uint8_t* feature = desc + (size_t)i * 32;
        int32_t nid = 0;  // root when nid_level <= 0
        int32_t final_id = 0;
        int current_level = 0;
        do {
            ++current_level;
            const int32_t* nodes = child_ids + child_begin[final_id];
            const int nn = child_count[final_id];
            final_id = nodes[0];
            double best_d = pso_hamming256(feature, node_desc + (size_t)final_id * 32);
            for (int c = 1; c < nn; ++c) {
                const int32_t id = nodes[c];
                const double d = pso_hamming256(feature, node_desc + (size_t)id * 32);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (child_count[final_id] != 0);
        const int32_t word_id = node_word[final_id];
        const double w = node_weight[final_id];
        f_word[i] = word_id; f_weight[i] = w; f_nid[i] = nid;
        if (w > 0) {
            auto it = v.lower_bound(word_id);
            if (it != v.end() && !(word_id < it->first)) it->second += w;
            else v.insert(it, std::make_pair(word_id, w));
            fv[nid].push_back(i);
        }
    }
    double norm = 0.0;  // L1_NORM: mustNormalize -> normalize(L1)
    for (auto& e : v) norm += std::fabs(e.second);
    if (norm > 0.0)
        for (auto& e : v) e.second /= norm;
    int k = 0;
    for (auto& e : v) { bow_id[k] = e.first; bow_val[k] = e.second; ++k; }
    int m = 0, p = 0;
    for (auto& e : fv) {
        fv_node[m] = e.first; fv_start[m] = p;
        for (int32_t i : e.second) fv_idx[p++] = i;
        ++m;
    }
    fv_start[m] = p;
    *n_fv = m;
    return k;
}

}  // extern "C"

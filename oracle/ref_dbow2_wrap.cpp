// oracle/_ref: Thirdparty/DBoW2/DBoW2/BowVector.cpp and FeatureVector.cpp - the code the reference links for Frame::ComputeBoW's
// accumulation (BowVector::addWeight, normalize(L1), FeatureVector::addFeature; row f2).  Built from the sources where they lie
// under /root/reference (oracle/Makefile target `_ref`; nothing of the reference is copied into this repository).  This wrapper
// is ours: one C entry point that feeds a stream of (word id, weight, node id) - what TemplatedVocabulary::transform produces per
// feature (TemplatedVocabulary.h:1165-1180) - through the reference's containers and flattens the result.  Test infrastructure only.
#include <cstdint>

#include "BowVector.h"
#include "FeatureVector.h"

// returns the number of BowVector entries; *n_fv = number of FeatureVector nodes; fv_start has *n_fv + 1 entries
extern "C" int ref_bow_accumulate(const int32_t* word, const double* weight, const int32_t* nid, int n, int32_t* bow_id, double* bow_val,
                                  int32_t* fv_node, int32_t* fv_start, int32_t* fv_idx, int* n_fv) {
    DBoW2::BowVector v;
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; ++i) {
        if (weight[i] > 0) {   // TemplatedVocabulary.h:1172-1178: stopped words (weight 0) are skipped
            v.addWeight((DBoW2::WordId)word[i], weight[i]);
            fv.addFeature((DBoW2::NodeId)nid[i], (unsigned int)i);
        }
    }
    v.normalize(DBoW2::L1);    // TemplatedVocabulary.h:1184: L1_NORM scoring must normalise
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { bow_id[k] = (int32_t)it->first; bow_val[k] = it->second; }
    int m = 0, p = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++m) {
        fv_node[m] = (int32_t)it->first;
        fv_start[m] = p;
        for (size_t j = 0; j < it->second.size(); ++j) fv_idx[p++] = (int32_t)it->second[j];
    }
    fv_start[m] = p;
    *n_fv = m;
    return k;
}

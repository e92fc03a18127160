// oracle/bow.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Parity unpinned: the reference holds no input/output
// vectors for DBoW2 or SearchByBoW; pinned here by brute-force definitional checks in tests/test_oracle_bow.py.
// CPU restatement of the DBoW2 vocabulary-tree descent (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1140-1272,
// BowVector.cpp:36-85, FeatureVector.cpp:31-45, FORB.cpp:80-101) as Frame::ComputeBoW calls it (src/Frame.cc:575-582,
// levelsup = 4), and of ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:159-288).
#pragma once
#include <stdint.h>
#include <vector>
#include <map>

namespace ora {

// Flat vocabulary tree: node 0 is the root; children of node n are child_ids[child_start[n] .. child_start[n+1]);
// a node without children is a leaf (a word) with word_id >= 0 and a TF-IDF weight.
struct Vocabulary {
    int n_nodes = 0, L = 0;
    const int32_t* child_start = nullptr; const int32_t* child_ids = nullptr;
    const uint8_t* desc = nullptr;            // [n_nodes][32]
    const int32_t* word_id = nullptr;         // [n_nodes]
    const double* weight = nullptr;           // [n_nodes]
};
// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
void bow_transform_one(const Vocabulary& V, const uint8_t* d, int levelsup, int& word, double& weight, int& node);
// transform(features, BowVector, FeatureVector, levelsup) with TF_IDF weighting and L1 scoring (ORBVocabulary)
void bow_transform(const Vocabulary& V, const uint8_t* desc, int n, int levelsup, std::map<int, double>& bow,
                   std::map<int, std::vector<unsigned>>& fv, int* word, double* weight, int* node);
// SearchByBoW: match[iF] = key-frame feature index or -1; returns nmatches
int search_by_bow(const std::map<int, std::vector<unsigned>>& fv_kf, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_has_point,
                  const std::map<int, std::vector<unsigned>>& fv_f, const uint8_t* f_desc, const float* f_angle, int nF,
                  float nnratio, bool check_orientation, std::vector<int>& match);

} // namespace ora

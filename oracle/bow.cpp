// oracle/bow.cpp — TEST INFRASTRUCTURE ONLY (see bow.h).
#include "bow.h"
#include "orb_matcher.h"
#include <cmath>
namespace ora {

void bow_transform_one(const Vocabulary& V, const uint8_t* d, int levelsup, int& word, double& weight, int& node) {
    const int nid_level = V.L - levelsup;                           // TemplatedVocabulary.h:1240-1241
    node = 0;
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const int c0 = V.child_start[final_id], c1 = V.child_start[final_id + 1];
        final_id = V.child_ids[c0];
        double best_d = (double)descriptor_distance(d, V.desc + (size_t)32 * final_id);
        for (int c = c0 + 1; c < c1; c++) {
            const int id = V.child_ids[c];
            const double dd = (double)descriptor_distance(d, V.desc + (size_t)32 * id);
            if (dd < best_d) { best_d = dd; final_id = id; }
        }
        if (current_level == nid_level) node = final_id;
    } while (V.child_start[final_id + 1] > V.child_start[final_id]);
    word = V.word_id[final_id]; weight = V.weight[final_id];
}

void bow_transform(const Vocabulary& V, const uint8_t* desc, int n, int levelsup, std::map<int, double>& bow,
                   std::map<int, std::vector<unsigned>>& fv, int* word, double* weight, int* node) {
    bow.clear(); fv.clear();
    if (V.n_nodes <= 1) return;
    for (int i = 0; i < n; i++) {
        int w, nd; double wt;
        bow_transform_one(V, desc + (size_t)32 * i, levelsup, w, wt, nd);
        if (word) word[i] = w;
        if (weight) weight[i] = wt;
        if (node) node[i] = nd;
        if (wt > 0) { bow[w] += wt; fv[nd].push_back((unsigned)i); }        // addWeight / addFeature
    }
    double norm = 0.0;                                               // BowVector::normalize(L1)
    for (auto& kv : bow) norm += std::fabs(kv.second);
    if (norm > 0.0) for (auto& kv : bow) kv.second /= norm;
}

int search_by_bow(const std::map<int, std::vector<unsigned>>& fv_kf, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_has_point,
                  const std::map<int, std::vector<unsigned>>& fv_f, const uint8_t* f_desc, const float* f_angle, int nF,
                  float nnratio, bool check_orientation, std::vector<int>& match) {
    match.assign(nF, -1);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    auto KFit = fv_kf.begin(); auto Fit = fv_f.begin();
    while (KFit != fv_kf.end() && Fit != fv_f.end()) {
        if (KFit->first == Fit->first) {
            for (unsigned realIdxKF : KFit->second) {
                if (!kf_has_point[realIdxKF]) continue;
                const uint8_t* dKF = kf_desc + (size_t)32 * realIdxKF;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (unsigned realIdxF : Fit->second) {
                    if (match[realIdxF] >= 0) continue;
                    const int dist = descriptor_distance(dKF, f_desc + (size_t)32 * realIdxF);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {
                    match[bestIdxF] = (int)realIdxKF;
                    if (check_orientation) {
                        float rot = kf_angle[realIdxKF] - f_angle[bestIdxF];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(bestIdxF);
                    }
                    nmatches++;
                }
            }
            ++KFit; ++Fit;
        } else if (KFit->first < Fit->first) KFit = fv_kf.lower_bound(Fit->first);
        else Fit = fv_f.lower_bound(KFit->first);
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { match[j] = -1; nmatches--; }
        }
    }
    return nmatches;
}
} // namespace ora

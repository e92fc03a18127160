// oracle/kf_matcher.cpp — TEST INFRASTRUCTURE ONLY (see kf_matcher.h).
#include "kf_matcher.h"
#include <cmath>
namespace ora {

// ORBmatcher.cc:138-156
static bool check_dist_epipolar_line(const KeyPoint& kp1, const KeyPoint& kp2, const float* F12, const float* level_sigma2_2) {
    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * level_sigma2_2[kp2.octave];            // double comparison, as written in the reference
}

int search_for_triangulation(const KfView& k1, const KfView& k2, const float* F12, const float* Cw1, const PoseF& pose2,
                             const float* sf2, const float* level_sigma2_2, bool only_stereo, bool check_orientation,
                             std::vector<int>& match12) {
    std::map<int, std::vector<unsigned>> fv1, fv2;
    for (int i = 0; i < k1.N; i++) if (k1.node[i] >= 0) fv1[k1.node[i]].push_back((unsigned)i);
    for (int i = 0; i < k2.N; i++) if (k2.node[i] >= 0) fv2[k2.node[i]].push_back((unsigned)i);
    float C2[3]; transform_point(pose2, Cw1, C2);                 // epipole in the second image (:663-670)
    const float invz = 1.0f / C2[2];
    const float ex = pose2.fx * C2[0] * invz + pose2.cx, ey = pose2.fy * C2[1] * invz + pose2.cy;
    int nmatches = 0;
    match12.assign(k1.N, -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    auto f1it = fv1.begin(); auto f2it = fv2.begin();
    while (f1it != fv1.end() && f2it != fv2.end()) {
        if (f1it->first == f2it->first) {
            for (unsigned idx1 : f1it->second) {
                if (k1.has_point[idx1]) continue;
                const bool bStereo1 = k1.uright[idx1] >= 0;
                if (only_stereo && !bStereo1) continue;
                const KeyPoint& kp1 = k1.kps[idx1];
                const uint8_t* d1 = k1.desc + (size_t)32 * idx1;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (unsigned idx2 : f2it->second) {
                    if (k2.has_point[idx2]) continue;             // (vbMatched2 is never set in the reference)
                    const bool bStereo2 = k2.uright[idx2] >= 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = descriptor_distance(d1, k2.desc + (size_t)32 * idx2);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const KeyPoint& kp2 = k2.kps[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kp2.x, distey = ey - kp2.y;
                        if (distex * distex + distey * distey < 100 * sf2[kp2.octave]) continue;
                    }
                    if (check_dist_epipolar_line(kp1, kp2, F12, level_sigma2_2)) { bestIdx2 = (int)idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2; nmatches++;
                    if (check_orientation) {
                        float rot = kp1.angle - k2.kps[bestIdx2].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back((int)idx1);
                    }
                }
            }
            ++f1it; ++f2it;
        } else if (f1it->first < f2it->first) f1it = fv1.lower_bound(f2it->first);
        else f2it = fv2.lower_bound(f1it->first);
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { match12[j] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int fuse(const FrameGrid& kf, const float* uright, const PoseF& T, float bf, const float* sf, const float* inv_level_sigma2, int nlevels,
         float log_scale_factor, const std::vector<FusePoint>& pts, float th, std::vector<int>& best_idx) {
    float Ow[3]; camera_centre(T, Ow);
    int nFused = 0;
    best_idx.assign(pts.size(), -1);
    for (size_t i = 0; i < pts.size(); i++) {
        const FusePoint& p = pts[i];
        if (!p.valid) continue;
        float Pc[3]; transform_point(T, p.Pw, Pc);
        if (Pc[2] < 0.0f) continue;
        const float invz = 1 / Pc[2];
        const float x = Pc[0] * invz, y = Pc[1] * invz;
        const float u = T.fx * x + T.cx, v = T.fy * y + T.cy;
        if (!(u >= kf.minX && u < kf.maxX && v >= kf.minY && v < kf.maxY)) continue;          // KeyFrame::IsInImage
        const float ur = u - bf * invz;
        const float maxDistance = 1.2f * p.max_dist, minDistance = 0.8f * p.min_dist;
        const float PO[3] = {p.Pw[0] - Ow[0], p.Pw[1] - Ow[1], p.Pw[2] - Ow[2]};
        const float dist3D = (float)std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        double dotp = 0; for (int k = 0; k < 3; k++) dotp += (double)PO[k] * p.normal[k];
        if (dotp < 0.5 * dist3D) continue;
        const float ratio = p.max_dist / dist3D;
        int nPredictedLevel = (int)std::ceil((float)std::log((double)ratio) / log_scale_factor);
        if (nPredictedLevel < 0) nPredictedLevel = 0; else if (nPredictedLevel >= nlevels) nPredictedLevel = nlevels - 1;
        const float radius = th * sf[nPredictedLevel];
        const std::vector<int> cand = kf.features_in_area(u, v, radius);                         // KeyFrame::GetFeaturesInArea: no level filter
        if (cand.empty()) continue;
        int bestDist = 256, bestIdx = -1;
        for (int idx : cand) {
            const KeyPoint& kp = kf.kps[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (uright[idx] >= 0) {
                const float ex = u - kp.x, ey = v - kp.y, er = ur - uright[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
            } else {
                const float ex = u - kp.x, ey = v - kp.y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
            }
            const int dist = descriptor_distance(p.desc, kf.desc + (size_t)32 * idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; nFused++; }
    }
    return nFused;
}
} // namespace ora

// oracle/orb_stereo.cpp — TEST INFRASTRUCTURE ONLY (see orb_stereo.h). Line references: src/Frame.cc.
#include "orb_stereo.h"
#include "orb_matcher.h"
#include <algorithm>
#include <climits>
#include <cmath>
namespace ora {
StereoResult compute_stereo_matches(const std::vector<KeyPoint>& keysL, const std::vector<uint8_t>& descL,
                                    const std::vector<KeyPoint>& keysR, const std::vector<uint8_t>& descR,
                                    const std::vector<Image8>& pyrL, const std::vector<Image8>& pyrR,
                                    const std::vector<float>& sf, const std::vector<float>& isf, float mbf, float fx) {
    const int N = (int)keysL.size(), Nr = (int)keysR.size();
    StereoResult R; R.uRight.assign(N, -1.0f); R.depth.assign(N, -1.0f); R.best_sad.assign(N, -1);
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = pyrL[0].h;
    std::vector<std::vector<int>> rows(nRows);                                // :653-673
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = keysR[iR].y, r = 2.0f * sf[keysR[iR].octave];
        const int maxr = (int)std::ceil(kpY + r), minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++) if (yi >= 0 && yi < nRows) rows[yi].push_back(iR);   // reference indexes unchecked
    }
    const float mb = mbf / fx, minZ = mb, minD = 0, maxD = mbf / minZ;        // :676-678 (mb = mbf/fx, :407)
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < N; iL++) {
        const KeyPoint& kpL = keysL[iL];
        const int levelL = kpL.octave; const float vL = kpL.y, uL = kpL.x;
        const int rowi = (int)vL;
        if (rowi < 0 || rowi >= nRows) continue;
        const std::vector<int>& cand = rows[rowi];
        if (cand.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH; int bestIdxR = 0;
        for (int iR : cand) {
            const KeyPoint& kpR = keysR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = descriptor_distance(&descL[(size_t)32 * iL], &descR[(size_t)32 * iR]);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {                                            // :732-803
            const float uR0 = keysR[bestIdxR].x;
            const float scaleFactor = isf[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor), scaledvL = std::round(kpL.y * scaleFactor), scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const Image8& IL = pyrL[kpL.octave]; const Image8& IRm = pyrR[kpL.octave];
            const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= IRm.w) continue;
            if (cv - w < 0 || cv + w >= IL.h || cu - w < 0 || cu + w >= IL.w || cr - L - w < 0 || cr + L + w >= IRm.w) continue;   // reference would assert
            int best = INT_MAX, bestincR = 0; float vDists[2 * 5 + 1];
            const float cL = (float)IL.at(cv, cu);
            for (int incR = -L; incR <= L; incR++) {
                const float cRv = (float)IRm.at(cv, cr + incR);
                double acc = 0;                                                // cv::norm(NORM_L1) on CV_32F accumulates in double
                for (int dy = -w; dy <= w; dy++) for (int dx = -w; dx <= w; dx++)
                    acc += std::fabs(((float)IL.at(cv + dy, cu + dx) - cL) - ((float)IRm.at(cv + dy, cr + incR + dx) - cRv));
                const float dist = (float)acc;
                if (dist < best) { best = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = sf[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }   // "bestuR = uL-0.01": a double subtraction rounded once (Frame.cc:797)
                R.depth[iL] = mbf / disparity; R.uRight[iL] = bestuR; R.best_sad[iL] = best;
                vDistIdx.push_back(std::make_pair(best, iL));
            }
        }
    }
    if (vDistIdx.empty()) return R;                                            // reference reads vDistIdx[0] unchecked
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        R.uRight[vDistIdx[i].second] = -1; R.depth[vDistIdx[i].second] = -1;
    }
    return R;
}
}

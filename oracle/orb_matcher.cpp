// oracle/orb_matcher.cpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md). PARITY UNPINNED for the
// float evaluation order of the cv::Mat expressions (OpenCV 2.4 gemm small-matrix path restated in
// transform_point); integer parts (Hamming, grid walk order, greedy ownership, histogram) follow the
// reference source line by line.
#include "orb_matcher.h"
#include <cmath>

namespace ora {

int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    const uint32_t* pa = reinterpret_cast<const uint32_t*>(a);
    const uint32_t* pb = reinterpret_cast<const uint32_t*>(b);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// Frame.cc:180-186 + AssignFeaturesToGrid :410-425
void FrameGrid::build(const KeyPoint* k, const uint8_t* d, int n, float min_x, float max_x, float min_y, float max_y) {
    N = n; kps = k; desc = d; minX = min_x; maxX = max_x; minY = min_y; maxY = max_y;
    wInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(maxX - minX);
    hInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(maxY - minY);
    for (int i = 0; i < FRAME_GRID_COLS; i++) for (int j = 0; j < FRAME_GRID_ROWS; j++) grid[i][j].clear();
    for (int i = 0; i < N; i++) { int px, py; if (pos_in_grid(kps[i], px, py)) grid[px][py].push_back(i); }
}
// Frame.cc:562-572 (note: round(), not floor())
bool FrameGrid::pos_in_grid(const KeyPoint& kp, int& px, int& py) const {
    px = (int)std::round((kp.x - minX) * wInv);
    py = (int)std::round((kp.y - minY) * hInv);
    return !(px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS);
}
// Frame.cc:507-560
std::vector<int> FrameGrid::features_in_area(float x, float y, float r, int minLevel, int maxLevel) const {
    std::vector<int> out;
    const int nMinCellX = std::max(0, (int)std::floor((x - minX - r) * wInv));
    if (nMinCellX >= FRAME_GRID_COLS) return out;
    const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)std::ceil((x - minX + r) * wInv));
    if (nMaxCellX < 0) return out;
    const int nMinCellY = std::max(0, (int)std::floor((y - minY - r) * hInv));
    if (nMinCellY >= FRAME_GRID_ROWS) return out;
    const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)std::ceil((y - minY + r) * hInv));
    if (nMaxCellY < 0) return out;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
            for (int idx : grid[ix][iy]) {
                const KeyPoint& kp = kps[idx];
                if (bCheckLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                const float distx = kp.x - x, disty = kp.y - y;
                if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(idx);
            }
    return out;
}

void transform_point(const PoseF& T, const float* X, float* Pc) {
    for (int r = 0; r < 3; r++) {
        const float t = T.Rcw[3 * r] * X[0] + T.Rcw[3 * r + 1] * X[1] + T.Rcw[3 * r + 2] * X[2];
        Pc[r] = (float)((double)t * 1.0 + (double)T.tcw[r] * 1.0);
    }
}

void compute_three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
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

// ORBmatcher.cc:1328-1471, monocular case (bForward = bBackward = false, mvuRight < 0 everywhere).
int search_by_projection_frame(const FrameGrid& cur, const PoseF& T, const float* sf,
                               const std::vector<LastFramePoint>& last, float th, bool check_ori,
                               std::vector<int>& cur_match) {
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<uint8_t> owner_has_obs(cur.N, 0);     // Observations() > 0 of the point currently assigned
    for (int i2 = 0; i2 < cur.N; i2++) owner_has_obs[i2] = (cur_match[i2] >= 0) ? last[cur_match[i2]].has_observations : 0;
    for (int i = 0; i < (int)last.size(); i++) {
        const LastFramePoint& lp = last[i];
        if (!lp.has_point || lp.outlier) continue;
        float Pc[3]; transform_point(T, lp.Pw, Pc);
        const float xc = Pc[0], yc = Pc[1];
        const float invzc = (float)(1.0 / Pc[2]);
        if (invzc < 0) continue;
        const float u = T.fx * xc * invzc + T.cx;
        const float v = T.fy * yc * invzc + T.cy;
        if (u < cur.minX || u > cur.maxX) continue;
        if (v < cur.minY || v > cur.maxY) continue;
        const int nLastOctave = lp.octave;
        const float radius = th * sf[nLastOctave];
        const std::vector<int> cand = cur.features_in_area(u, v, radius, nLastOctave - 1, nLastOctave + 1);
        if (cand.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : cand) {
            if (cur_match[i2] >= 0 && owner_has_obs[i2]) continue;
            const int dist = descriptor_distance(lp.desc, cur.desc + (size_t)32 * i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            cur_match[bestIdx2] = i; owner_has_obs[bestIdx2] = lp.has_observations;
            nmatches++;
            if (check_ori) {
                float rot = lp.angle - cur.kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_ori) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { cur_match[idx] = -1; nmatches--; }
    }
    return nmatches;
}

} // namespace ora

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

void camera_centre(const PoseF& T, float* Ow);
// ORBmatcher.cc:1328-1471. stereo == nullptr: the monocular case (bForward = bBackward = false, mvuRight < 0 everywhere).
int search_by_projection_frame(const FrameGrid& cur, const PoseF& T, const float* sf,
                               const std::vector<LastFramePoint>& last, float th, bool check_ori,
                               std::vector<int>& cur_match, const StereoSearch* stereo) {
    int nmatches = 0;
    bool bForward = false, bBackward = false;
    if (stereo) {
        // twc = -Rcw.t()*tcw ; tlc = Rlw*twc+tlw (:1339-1346), 3-term float sums as OpenCV's small-matrix gemm forms them
        float twc[3], tlc[3];
        camera_centre(T, twc);
        for (int r = 0; r < 3; r++) {
            const float t = stereo->last.Rcw[3 * r] * twc[0] + stereo->last.Rcw[3 * r + 1] * twc[1] + stereo->last.Rcw[3 * r + 2] * twc[2];
            tlc[r] = t + stereo->last.tcw[r];
        }
        bForward = tlc[2] > stereo->mb; bBackward = -tlc[2] > stereo->mb;
    }
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
        const std::vector<int> cand = bForward ? cur.features_in_area(u, v, radius, nLastOctave, -1)
                                    : (bBackward ? cur.features_in_area(u, v, radius, 0, nLastOctave)
                                                 : cur.features_in_area(u, v, radius, nLastOctave - 1, nLastOctave + 1));
        if (cand.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : cand) {
            if (cur_match[i2] >= 0 && owner_has_obs[i2]) continue;
            if (stereo && stereo->uright[i2] > 0) {
                const float ur = u - stereo->bf * invzc;
                const float er = std::fabs(ur - stereo->uright[i2]);
                if (er > radius) continue;
            }
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

void camera_centre(const PoseF& T, float* Ow) {
    // mOw = -mRcw.t()*mtcw : gemm with alpha = -1 on the 3-term float sums of the transposed rows
    for (int r = 0; r < 3; r++) {
        const float t = T.Rcw[r] * T.tcw[0] + T.Rcw[3 + r] * T.tcw[1] + T.Rcw[6 + r] * T.tcw[2];
        Ow[r] = -t;
    }
}

// Frame.cc:449-505
FrustumResult is_in_frustum(const PoseF& T, const float* Ow, float min_x, float max_x, float min_y, float max_y,
                            float log_scale_factor, int nlevels, const LocalPoint& p, float viewingCosLimit, float bf) {
    FrustumResult R{0, 0, 0, 0, 0, 0};
    float Pc[3]; transform_point(T, p.Pw, Pc);
    const float PcX = Pc[0], PcY = Pc[1], PcZ = Pc[2];
    if (PcZ < 0.0f) return R;
    const float invz = 1.0f / PcZ;
    const float u = T.fx * PcX * invz + T.cx;
    const float v = T.fy * PcY * invz + T.cy;
    if (u < min_x || u > max_x) return R;
    if (v < min_y || v > max_y) return R;
    const float maxDistance = 1.2f * p.max_dist, minDistance = 0.8f * p.min_dist;
    const float PO[3] = {p.Pw[0] - Ow[0], p.Pw[1] - Ow[1], p.Pw[2] - Ow[2]};
    // cv::norm(NORM_L2) on CV_32F accumulates in double; Mat::dot accumulates in double
    const float dist = (float)std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
    if (dist < minDistance || dist > maxDistance) return R;
    double dotp = 0; for (int i = 0; i < 3; i++) dotp += (double)PO[i] * p.normal[i];
    const float viewCos = (float)(dotp / dist);
    if (viewCos < viewingCosLimit) return R;
    // MapPoint::PredictScale: the reference's log(ratio) is libm logf (not correctly rounded in glibc); here the
    // double log rounded once to float, like cos/sin in the descriptor (DESIGN.md deviations). Only feeds ceil().
    const float ratio = p.max_dist / dist;
    int nScale = (int)std::ceil((float)std::log((double)ratio) / log_scale_factor);
    if (nScale < 0) nScale = 0; else if (nScale >= nlevels) nScale = nlevels - 1;
    R.in_view = 1; R.proj_x = u; R.proj_y = v; R.level = nScale; R.view_cos = viewCos;
    R.proj_xr = u - bf * invz;                                   // pMP->mTrackProjXR = u - mbf*invz (Frame.cc:499), float
    return R;
}

// Tracking.cc:1922-1937 + ORBmatcher.cc:45-129
int search_local_points(const FrameGrid& cur, const PoseF& T, const float* sf, int nlevels, float log_scale_factor,
                        const std::vector<LocalPoint>& pts, float th, float nnratio, const uint8_t* cur_owner_obs,
                        std::vector<int>& match, std::vector<FrustumResult>* frustum, const float* cur_uright, float bf) {
    float Ow[3]; camera_centre(T, Ow);
    int nmatches = 0;
    const bool bFactor = th != 1.0f;
    match.assign(cur.N, -1);
    std::vector<uint8_t> owner_obs(cur_owner_obs, cur_owner_obs + cur.N);
    if (frustum) frustum->assign(pts.size(), FrustumResult{0, 0, 0, 0, 0, 0});
    for (size_t i = 0; i < pts.size(); i++) {
        const LocalPoint& p = pts[i];
        if (p.skip || !p.valid) continue;                         // mnLastFrameSeen == id / isBad(): isInFrustum is not called
        const FrustumResult fr = is_in_frustum(T, Ow, cur.minX, cur.maxX, cur.minY, cur.maxY, log_scale_factor, nlevels, p, 0.5f, bf);
        if (frustum) (*frustum)[i] = fr;
        if (!fr.in_view) continue;
        float r = fr.view_cos > 0.998 ? 2.5f : 4.0f;              // RadiusByViewingCos
        if (bFactor) r *= th;
        const std::vector<int> cand = cur.features_in_area(fr.proj_x, fr.proj_y, r * sf[fr.level], fr.level - 1, fr.level);
        if (cand.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : cand) {
            if (owner_obs[idx]) continue;
            if (cur_uright && cur_uright[idx] > 0) {               // ORBmatcher.cc:91-97
                const float er = std::fabs(fr.proj_xr - cur_uright[idx]);
                if (er > r * sf[fr.level]) continue;
            }
            const int dist = descriptor_distance(p.desc, cur.desc + (size_t)32 * idx);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = cur.kps[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = cur.kps[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            match[bestIdx] = (int)i; owner_obs[bestIdx] = p.has_observations;
            nmatches++;
        }
    }
    return nmatches;
}

} // namespace ora

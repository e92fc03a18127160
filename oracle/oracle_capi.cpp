// oracle/oracle_capi.cpp — TEST INFRASTRUCTURE ONLY. extern "C" surface of the CPU oracle so that
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it through ctypes.
// Nothing under viorb_amd/ may link, import or call this library.
#include "orb_extractor.h"
#include <cstring>

using namespace ora;

extern "C" {

void* ora_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    return new OrbExtractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void ora_extractor_destroy(void* h) { delete (OrbExtractor*)h; }

// Returns the number of keypoints (also when it exceeds cap; only min(n,cap) are written).
int ora_extract(void* h, const uint8_t* img, int w, int hgt, int stride, KeyPoint* kps,
                uint8_t* desc, int cap) {
    OrbExtractor* e = (OrbExtractor*)h;
    std::vector<KeyPoint> k; std::vector<uint8_t> d;
    int n = e->extract(img, w, hgt, stride, k, d);
    int m = n < cap ? n : cap;
    if (m > 0) { std::memcpy(kps, k.data(), sizeof(KeyPoint) * m); std::memcpy(desc, d.data(), (size_t)32 * m); }
    return n;
}
void ora_extractor_tables(void* h, float* sf, float* isf, float* s2, float* is2, int* quota, int* umax16) {
    OrbExtractor* e = (OrbExtractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        sf[i] = e->mvScaleFactor[i]; isf[i] = e->mvInvScaleFactor[i];
        s2[i] = e->mvLevelSigma2[i]; is2[i] = e->mvInvLevelSigma2[i];
        quota[i] = e->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}
int ora_level_size(void* h, int l, int* w, int* hgt) {
    OrbExtractor* e = (OrbExtractor*)h;
    if (l < 0 || l >= (int)e->pyramid.size()) return -1;
    *w = e->pyramid[l].w; *hgt = e->pyramid[l].h; return 0;
}
// which: 0 = pyramid level, 1 = blurred level. dst must hold w*h bytes.
int ora_level_copy(void* h, int l, int which, uint8_t* dst) {
    OrbExtractor* e = (OrbExtractor*)h;
    const std::vector<Image8>& v = which ? e->blurred : e->pyramid;
    if (l < 0 || l >= (int)v.size()) return -1;
    std::memcpy(dst, v[l].d.data(), v[l].d.size());
    return (int)v[l].d.size();
}
// which: 0 = FAST candidates (pre-octree, relative to the 16-px border origin), 1 = level keypoints.
int ora_level_keypoints(void* h, int l, int which, KeyPoint* dst, int cap) {
    OrbExtractor* e = (OrbExtractor*)h;
    const std::vector<std::vector<KeyPoint>>& v = which ? e->level_kps : e->candidates;
    if (l < 0 || l >= (int)v.size()) return -1;
    int n = (int)v[l].size(), m = n < cap ? n : cap;
    if (m > 0) std::memcpy(dst, v[l].data(), sizeof(KeyPoint) * m);
    return n;
}

// --- primitives, exposed one by one for the definitional tests ---------------------------------
void ora_resize_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    Image8 s(sw, sh); std::memcpy(s.d.data(), src, (size_t)sw * sh);
    Image8 d; resize_linear_8u(s, d, dw, dh);
    std::memcpy(dst, d.d.data(), (size_t)dw * dh);
}
void ora_gaussian_blur(const uint8_t* src, int w, int h, uint8_t* dst) {
    Image8 s(w, h); std::memcpy(s.d.data(), src, (size_t)w * h);
    Image8 d; gaussian_blur_7x7_s2(s, d);
    std::memcpy(dst, d.d.data(), (size_t)w * h);
}
void ora_gaussian_kernel_q8(int n, double sigma, int* k) { gaussian_kernel_q8(n, sigma, k); }
float ora_fast_atan2(float y, float x) { return fastAtan2(y, x); }
int ora_cv_round(double v) { return cvRound(v); }
// FAST on the sub-image [x0,x1) x [y0,y1); out = (x,y,score) int triples; returns count.
int ora_fast(const uint8_t* img, int w, int h, int x0, int y0, int x1, int y1, int threshold,
             int* out, int cap) {
    Image8 s(w, h); std::memcpy(s.d.data(), img, (size_t)w * h);
    std::vector<FastKP> v; fast9_16(s, x0, y0, x1, y1, threshold, v);
    int n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++) { out[3 * i] = v[i].x; out[3 * i + 1] = v[i].y; out[3 * i + 2] = v[i].score; }
    return n;
}
float ora_ic_angle(const uint8_t* img, int w, int h, float x, float y) {
    Image8 s(w, h); std::memcpy(s.d.data(), img, (size_t)w * h);
    OrbExtractor e(1000, 1.2f, 8, 20, 7);
    return ic_angle(s, x, y, e.umax);
}
void ora_orb_descriptor(const uint8_t* blurred, int w, int h, float x, float y, float angle, uint8_t* desc) {
    Image8 s(w, h); std::memcpy(s.d.data(), blurred, (size_t)w * h);
    OrbExtractor e(1000, 1.2f, 8, 20, 7);
    KeyPoint kp{x, y, 31.f, angle, 0.f, 0, -1};
    orb_descriptor(kp, s, e.pattern, desc);
}
int ora_distribute_octree(const KeyPoint* keys, int n, int minX, int maxX, int minY, int maxY, int N,
                          KeyPoint* out, int cap) {
    OrbExtractor e(1000, 1.2f, 8, 20, 7);
    std::vector<KeyPoint> v(keys, keys + n);
    std::vector<KeyPoint> r = e.distribute_octree(v, minX, maxX, minY, maxY, N);
    int m = (int)r.size();
    for (int i = 0; i < m && i < cap; i++) out[i] = r[i];
    return m;
}

} // extern "C"

// =================================================================================================
// Visual-inertial part (vio.h). Flat double layouts shared with include/viorb.h:
//   navstate[22] = P3 V3 q4(x,y,z,w) bg3 ba3 dbg3 dba3
//   preint[142]  = dP3 dV3 dR9 JPg9 JPa9 JVg9 JVa9 JRg9 cov81 dt        (matrices row-major)
//   cam[16]      = fx fy cx cy Rbc9 Pbc3
//   obs[n][6]    = Pw3 u v invSigma2
// =================================================================================================
#include "vio.h"
namespace {
V3 v3(const double* p) { return V3{p[0], p[1], p[2]}; }
void put3(double* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
M3 m3(const double* p) { M3 m; for (int i = 0; i < 9; i++) m.m[i] = p[i]; return m; }
void putm3(double* p, const M3& m) { for (int i = 0; i < 9; i++) p[i] = m.m[i]; }
NavState ns_in(const double* p) {
    NavState n; n.P = v3(p); n.V = v3(p + 3); n.R = SO3(Quat{p[6], p[7], p[8], p[9]});
    n.bg = v3(p + 10); n.ba = v3(p + 13); n.dbg = v3(p + 16); n.dba = v3(p + 19); return n;
}
void ns_out(double* p, const NavState& n) {
    put3(p, n.P); put3(p + 3, n.V); p[6] = n.R.q.x; p[7] = n.R.q.y; p[8] = n.R.q.z; p[9] = n.R.q.w;
    put3(p + 10, n.bg); put3(p + 13, n.ba); put3(p + 16, n.dbg); put3(p + 19, n.dba);
}
Preint pre_in(const double* p) {
    Preint M; M.dP = v3(p); M.dV = v3(p + 3); M.dR = m3(p + 6); M.JPg = m3(p + 15); M.JPa = m3(p + 24);
    M.JVg = m3(p + 33); M.JVa = m3(p + 42); M.JRg = m3(p + 51);
    for (int i = 0; i < 81; i++) M.cov.a[i] = p[60 + i];
    M.dt = p[141]; return M;
}
void pre_out(double* p, const Preint& M) {
    put3(p, M.dP); put3(p + 3, M.dV); putm3(p + 6, M.dR); putm3(p + 15, M.JPg); putm3(p + 24, M.JPa);
    putm3(p + 33, M.JVg); putm3(p + 42, M.JVa); putm3(p + 51, M.JRg);
    for (int i = 0; i < 81; i++) p[60 + i] = M.cov.a[i];
    p[141] = M.dt;
}
Camera cam_in(const double* p) { Camera c; c.fx = p[0]; c.fy = p[1]; c.cx = p[2]; c.cy = p[3]; c.Rbc = m3(p + 4); c.Pbc = v3(p + 13); return c; }
std::vector<Observation> obs_in(const double* p, int n) {
    std::vector<Observation> o(n);
    for (int i = 0; i < n; i++) { o[i].Pw = v3(p + 6 * i); o[i].u = p[6 * i + 3]; o[i].v = p[6 * i + 4]; o[i].inv_sigma2 = p[6 * i + 5]; }
    return o;
}
void mat_out(double* p, const Mat& m) { for (size_t i = 0; i < m.a.size(); i++) p[i] = m.a[i]; }
void result_out(const PoseOptResult& R, double* ns_cur, double* ns_last, uint8_t* oc, uint8_t* ol, double* marg144,
                double* info4, double* trace, int trace_cap) {
    ns_out(ns_cur, R.ns); if (ns_last) ns_out(ns_last, R.ns_last);
    for (size_t i = 0; i < R.outlier_cur.size(); i++) oc[i] = R.outlier_cur[i];
    if (ol) for (size_t i = 0; i < R.outlier_last.size(); i++) ol[i] = R.outlier_last[i];
    if (marg144 && R.marg_cov_inv.r == 12) mat_out(marg144, R.marg_cov_inv);
    info4[0] = R.n_inliers; info4[1] = R.final_chi2; info4[2] = R.lm_iterations; info4[3] = (double)R.chi2_trace.size();
    for (int i = 0; i < (int)R.chi2_trace.size() && i < trace_cap; i++) trace[i] = R.chi2_trace[i];
}
} // namespace

extern "C" {

void ora_pose_opt_diagnostics(int* out2) { pose_opt_diagnostics(out2, out2 + 1); }

void ora_preintegrate(const double* samples7, int n, const double* bg, const double* ba, double t_last, double t_cur, double* out142) {
    std::vector<ImuSample> s(n);
    for (int i = 0; i < n; i++) { for (int k = 0; k < 3; k++) { s[i].g[k] = samples7[7 * i + k]; s[i].a[k] = samples7[7 * i + 3 + k]; } s[i].t = samples7[7 * i + 6]; }
    Preint M; preintegrate(s.data(), n, v3(bg), v3(ba), t_last, t_cur, M);
    pre_out(out142, M);
}
void ora_preint_update(double* preint142, const double* omega, const double* acc, double dt) {
    Preint M = pre_in(preint142); M.update(v3(omega), v3(acc), dt); pre_out(preint142, M);
}
void ora_update_ns(double* ns22, const double* preint142, const double* gw) {
    NavState n = ns_in(ns22); update_ns(n, pre_in(preint142), v3(gw)); ns_out(ns22, n);
}
void ora_predict_navstate(const double* last22, const double* preint142, const double* gw, double* out22) {
    ns_out(out22, predict_navstate(ns_in(last22), pre_in(preint142), v3(gw)));
}
void ora_pose_from_navstate(const double* ns22, const double* cam16, float* pose12) {
    pose_from_navstate_f32(ns_in(ns22), cam_in(cam16), pose12);
}
void ora_ns_inc_pvr(double* ns22, const double* u9) { NavState n = ns_in(ns22); n.inc_small_pvr(u9); ns_out(ns22, n); }
void ora_so3_exp(const double* w, double* q4) { SO3 r = SO3::exp(v3(w)); q4[0] = r.q.x; q4[1] = r.q.y; q4[2] = r.q.z; q4[3] = r.q.w; }
void ora_so3_log(const double* q4, double* w) { put3(w, SO3(Quat{q4[0], q4[1], q4[2], q4[3]}).log()); }
void ora_so3_matrix(const double* q4, double* R9) { putm3(R9, SO3(Quat{q4[0], q4[1], q4[2], q4[3]}).matrix()); }
void ora_so3_from_matrix(const double* R9, double* q4) { SO3 r(m3(R9)); q4[0] = r.q.x; q4[1] = r.q.y; q4[2] = r.q.z; q4[3] = r.q.w; }
void ora_jacobian_r(const double* w, double* J9, int inverse) { putm3(J9, inverse ? jacobian_r_inv(v3(w)) : jacobian_r(v3(w))); }

void ora_edge_pvr(const double* i22, const double* j22, const double* b22, const double* preint142, const double* gw,
                  double* e9, double* Ji81, double* Jj81, double* Jb27) {
    NavState ni = ns_in(i22), nj = ns_in(j22), nb = ns_in(b22); Preint M = pre_in(preint142);
    edge_pvr_error(ni, nj, nb, M, v3(gw), e9);
    if (Ji81) { Mat Ji, Jj, Jb; edge_pvr_jacobians(ni, nj, nb, M, v3(gw), e9, Ji, Jj, Jb); mat_out(Ji81, Ji); mat_out(Jj81, Jj); mat_out(Jb27, Jb); }
}
void ora_edge_proj(const double* ns22, const double* cam16, const double* obs6, double* e2, double* J18) {
    NavState n = ns_in(ns22); Camera c = cam_in(cam16); Observation o = obs_in(obs6, 1)[0];
    edge_proj_error(n, c, o, e2);
    if (J18) { Mat J; edge_proj_jacobian(n, c, o, J); mat_out(J18, J); }
}
void ora_edge_prior(const double* pvr22, const double* bias22, const double* prior22, double* e12, double* Jp108, double* Jb36) {
    NavState p = ns_in(pvr22), b = ns_in(bias22), pr = ns_in(prior22);
    edge_prior_error(p, b, pr, e12);
    if (Jp108) { Mat Jp, Jb; edge_prior_jacobians(p, e12, Jp, Jb); mat_out(Jp108, Jp); mat_out(Jb36, Jb); }
}
void ora_pose_opt_vi_kf(const double* cur22, const double* kf22, const double* preint142, const double* gw, const double* cam16,
                        const double* obs6, int n, int marg, double* out_ns22, uint8_t* outlier, double* marg144,
                        double* info4, double* trace, int trace_cap) {
    PoseOptResult R = pose_opt_vi_kf(ns_in(cur22), ns_in(kf22), pre_in(preint142), v3(gw), cam_in(cam16), obs_in(obs6, n), marg != 0);
    result_out(R, out_ns22, nullptr, outlier, nullptr, marg144, info4, trace, trace_cap);
}
void ora_pose_opt_vi_frame(const double* cur22, const double* last22, const double* prior22, const double* margcovinv144,
                           const double* preint142, const double* gw, const double* cam16, const double* obs_cur6, int ncur,
                           const double* obs_last6, int nlast, int marg, double* out_ns22, double* out_last22,
                           uint8_t* outlier_cur, uint8_t* outlier_last, double* marg144, double* info4, double* trace, int trace_cap) {
    Mat mci(12, 12); for (int i = 0; i < 144; i++) mci.a[i] = margcovinv144[i];
    PoseOptResult R = pose_opt_vi_frame(ns_in(cur22), ns_in(last22), ns_in(prior22), mci, pre_in(preint142), v3(gw), cam_in(cam16),
                                        obs_in(obs_cur6, ncur), obs_in(obs_last6, nlast), marg != 0);
    result_out(R, out_ns22, out_last22, outlier_cur, outlier_last, marg144, info4, trace, trace_cap);
}

} // extern "C"

// =================================================================================================
// Matcher (orb_matcher.h)
// =================================================================================================
#include "orb_matcher.h"
extern "C" {
int ora_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }
// The bestDist1 / bestDist2 / bestIdx scan of the reference's searches (src/ORBmatcher.cc:204-222: strict '<', both distances start
// at 256) over ALL candidates: the brute-force matcher north_star names.
void ora_match_bruteforce(const uint8_t* q, int nq, const uint8_t* c, int nc, int* best, int* second, int* idx) {
    for (int i = 0; i < nq; i++) {
        int bestDist1 = 256, bestIdx = -1, bestDist2 = 256;
        for (int j = 0; j < nc; j++) {
            const int dist = descriptor_distance(q + (size_t)32 * i, c + (size_t)32 * j);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx = j; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        best[i] = bestDist1; second[i] = bestDist2; idx[i] = bestIdx;
    }
}
// Grid introspection: CSR of the 64x48 grid in the reference's storage order grid[ix][iy]
// (cell index = ix*48 + iy): cell_start[64*48+1], cell_idx[N]. Returns number of keypoints binned.
int ora_frame_grid(const KeyPoint* kps, int n, float minX, float maxX, float minY, float maxY, int* cell_start, int* cell_idx) {
    FrameGrid g; g.build(kps, nullptr, n, minX, maxX, minY, maxY);
    int pos = 0;
    for (int ix = 0; ix < FRAME_GRID_COLS; ix++) for (int iy = 0; iy < FRAME_GRID_ROWS; iy++) {
        cell_start[ix * FRAME_GRID_ROWS + iy] = pos;
        for (int idx : g.grid[ix][iy]) cell_idx[pos++] = idx;
    }
    cell_start[FRAME_GRID_COLS * FRAME_GRID_ROWS] = pos;
    return pos;
}
int ora_features_in_area(const KeyPoint* kps, int n, float minX, float maxX, float minY, float maxY, float x, float y, float r,
                         int minLevel, int maxLevel, int* out, int cap) {
    FrameGrid g; g.build(kps, nullptr, n, minX, maxX, minY, maxY);
    std::vector<int> v = g.features_in_area(x, y, r, minLevel, maxLevel);
    for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
    return (int)v.size();
}
// pose12 = Rcw (9, row-major) + tcw (3); intr = fx fy cx cy; bounds = minX maxX minY maxY.
// last_flags[i] = bit0 has_point | bit1 outlier | bit2 has_observations.
int ora_search_by_projection_frame(const KeyPoint* cur_kps, const uint8_t* cur_desc, int ncur, const float* bounds4,
                                   const float* pose12, const float* intr4, const float* scale_factors,
                                   int nlast, const uint8_t* last_flags, const float* last_Pw, const uint8_t* last_mp_desc,
                                   const int* last_octave, const float* last_angle, float th, int check_ori, int* cur_match) {
    FrameGrid g; g.build(cur_kps, cur_desc, ncur, bounds4[0], bounds4[1], bounds4[2], bounds4[3]);
    PoseF T; for (int i = 0; i < 9; i++) T.Rcw[i] = pose12[i]; for (int i = 0; i < 3; i++) T.tcw[i] = pose12[9 + i];
    T.fx = intr4[0]; T.fy = intr4[1]; T.cx = intr4[2]; T.cy = intr4[3];
    std::vector<LastFramePoint> last(nlast);
    for (int i = 0; i < nlast; i++) {
        last[i].has_point = last_flags[i] & 1; last[i].outlier = (last_flags[i] >> 1) & 1; last[i].has_observations = (last_flags[i] >> 2) & 1;
        for (int k = 0; k < 3; k++) last[i].Pw[k] = last_Pw[3 * i + k];
        last[i].desc = last_mp_desc + (size_t)32 * i; last[i].octave = last_octave[i]; last[i].angle = last_angle[i];
    }
    std::vector<int> m(cur_match, cur_match + ncur);
    int n = search_by_projection_frame(g, T, scale_factors, last, th, check_ori != 0, m);
    for (int i = 0; i < ncur; i++) cur_match[i] = m[i];
    return n;
}
// bMono == false: + CurrentFrame.mvuRight, LastFrame.mTcw, mbf, mb
int ora_search_by_projection_frame_stereo(const KeyPoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int ncur, const float* bounds4,
                                          const float* pose12, const float* last_pose12, const float* intr4, float bf, float mb,
                                          const float* scale_factors, int nlast, const uint8_t* last_flags, const float* last_Pw,
                                          const uint8_t* last_mp_desc, const int* last_octave, const float* last_angle, float th, int check_ori,
                                          int* cur_match) {
    FrameGrid g; g.build(cur_kps, cur_desc, ncur, bounds4[0], bounds4[1], bounds4[2], bounds4[3]);
    PoseF T; for (int i = 0; i < 9; i++) T.Rcw[i] = pose12[i]; for (int i = 0; i < 3; i++) T.tcw[i] = pose12[9 + i];
    T.fx = intr4[0]; T.fy = intr4[1]; T.cx = intr4[2]; T.cy = intr4[3];
    StereoSearch st; st.mb = mb; st.bf = bf; st.uright = cur_uright; st.last = T;
    for (int i = 0; i < 9; i++) st.last.Rcw[i] = last_pose12[i];
    for (int i = 0; i < 3; i++) st.last.tcw[i] = last_pose12[9 + i];
    std::vector<LastFramePoint> last(nlast);
    for (int i = 0; i < nlast; i++) {
        last[i].has_point = last_flags[i] & 1; last[i].outlier = (last_flags[i] >> 1) & 1; last[i].has_observations = (last_flags[i] >> 2) & 1;
        for (int k = 0; k < 3; k++) last[i].Pw[k] = last_Pw[3 * i + k];
        last[i].desc = last_mp_desc + (size_t)32 * i; last[i].octave = last_octave[i]; last[i].angle = last_angle[i];
    }
    std::vector<int> m(cur_match, cur_match + ncur);
    int n = search_by_projection_frame(g, T, scale_factors, last, th, check_ori != 0, m, &st);
    for (int i = 0; i < ncur; i++) cur_match[i] = m[i];
    return n;
}
} // extern "C"

extern "C" {
// pts_f[n][8] = Pw3 normal3 minDist maxDist; pts_flags[n] bit0 valid, bit1 skip, bit2 has observations.
// frustum_out[n][5] = in_view projx projy viewcos level (may be NULL).
static int search_local_points_impl(const float* cur_uright, float bf, float* proj_xr_out, const KeyPoint* cur_kps, const uint8_t* cur_desc, int ncur, const float* bounds4, const float* pose12,
                            const float* intr4, const float* scale_factors, int nlevels, float log_scale_factor, int npts,
                            const float* pts_f, const uint8_t* pts_flags, const uint8_t* pts_desc, float th, float nnratio,
                            const uint8_t* cur_owner_obs, int* match, float* frustum_out) {
    FrameGrid g; g.build(cur_kps, cur_desc, ncur, bounds4[0], bounds4[1], bounds4[2], bounds4[3]);
    PoseF T; for (int i = 0; i < 9; i++) T.Rcw[i] = pose12[i]; for (int i = 0; i < 3; i++) T.tcw[i] = pose12[9 + i];
    T.fx = intr4[0]; T.fy = intr4[1]; T.cx = intr4[2]; T.cy = intr4[3];
    std::vector<LocalPoint> pts(npts);
    for (int i = 0; i < npts; i++) {
        pts[i].valid = pts_flags[i] & 1; pts[i].skip = (pts_flags[i] >> 1) & 1; pts[i].has_observations = (pts_flags[i] >> 2) & 1;
        for (int k = 0; k < 3; k++) { pts[i].Pw[k] = pts_f[8 * i + k]; pts[i].normal[k] = pts_f[8 * i + 3 + k]; }
        pts[i].min_dist = pts_f[8 * i + 6]; pts[i].max_dist = pts_f[8 * i + 7]; pts[i].desc = pts_desc + (size_t)32 * i;
    }
    std::vector<int> m; std::vector<FrustumResult> fr;
    int n = search_local_points(g, T, scale_factors, nlevels, log_scale_factor, pts, th, nnratio, cur_owner_obs, m, &fr, cur_uright, bf);
    if (proj_xr_out) for (int i = 0; i < npts; i++) proj_xr_out[i] = fr[i].proj_xr;
    for (int i = 0; i < ncur; i++) match[i] = m[i];
    if (frustum_out) for (int i = 0; i < npts; i++) {
        frustum_out[5 * i] = fr[i].in_view; frustum_out[5 * i + 1] = fr[i].proj_x; frustum_out[5 * i + 2] = fr[i].proj_y;
        frustum_out[5 * i + 3] = fr[i].view_cos; frustum_out[5 * i + 4] = (float)fr[i].level;
    }
    return n;
}
int ora_search_local_points(const KeyPoint* cur_kps, const uint8_t* cur_desc, int ncur, const float* bounds4, const float* pose12,
                            const float* intr4, const float* scale_factors, int nlevels, float log_scale_factor, int npts,
                            const float* pts_f, const uint8_t* pts_flags, const uint8_t* pts_desc, float th, float nnratio,
                            const uint8_t* cur_owner_obs, int* match, float* frustum_out) {
    return search_local_points_impl(nullptr, 0.0f, nullptr, cur_kps, cur_desc, ncur, bounds4, pose12, intr4, scale_factors, nlevels, log_scale_factor, npts, pts_f, pts_flags,
                                    pts_desc, th, nnratio, cur_owner_obs, match, frustum_out);
}
// the same with a stereo / RGB-D current frame: cur_uright = mvuRight, bf = mbf; proj_xr_out[n] = mTrackProjXR (may be NULL)
int ora_search_local_points_stereo(const KeyPoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, float bf, int ncur, const float* bounds4,
                                   const float* pose12, const float* intr4, const float* scale_factors, int nlevels, float log_scale_factor, int npts,
                                   const float* pts_f, const uint8_t* pts_flags, const uint8_t* pts_desc, float th, float nnratio,
                                   const uint8_t* cur_owner_obs, int* match, float* frustum_out, float* proj_xr_out) {
    return search_local_points_impl(cur_uright, bf, proj_xr_out, cur_kps, cur_desc, ncur, bounds4, pose12, intr4, scale_factors, nlevels, log_scale_factor, npts, pts_f,
                                    pts_flags, pts_desc, th, nnratio, cur_owner_obs, match, frustum_out);
}
} // extern "C"

extern "C" {
// obs7[n][7] = Xw3 u v ur invSigma2 (ur < 0: mono); intr5 = fx fy cx cy bf; info3 = inliers, chi2, LM iterations.
void ora_pose_opt_se3(const float* pose12, const double* intr5, const double* obs7, int n, float* out_pose12, uint8_t* outlier, double* info3) {
    std::vector<Se3Obs> o(n);
    for (int i = 0; i < n; i++) { o[i].Xw = v3(obs7 + 7 * i); o[i].u = obs7[7 * i + 3]; o[i].v = obs7[7 * i + 4]; o[i].ur = obs7[7 * i + 5]; o[i].inv_sigma2 = obs7[7 * i + 6]; }
    Se3Result R = pose_opt_se3(pose12, intr5[0], intr5[1], intr5[2], intr5[3], intr5[4], o);
    for (int i = 0; i < 12; i++) out_pose12[i] = R.pose12[i];
    for (int i = 0; i < n; i++) outlier[i] = R.outlier[i];
    info3[0] = R.n_inliers; info3[1] = R.final_chi2; info3[2] = R.lm_iterations;
}
} // extern "C"

#include "orb_stereo.h"
extern "C" {
// Stereo matching on the features / pyramids two oracle extractors hold after extract() (left, right).
int ora_stereo_match(void* hl, void* hr, const KeyPoint* kl, const uint8_t* dl, int nl, const KeyPoint* kr, const uint8_t* dr, int nr,
                     float bf, float fx, float* uRight, float* depth, int* best_sad) {
    OrbExtractor* L = (OrbExtractor*)hl; OrbExtractor* Rr = (OrbExtractor*)hr;
    std::vector<KeyPoint> KL(kl, kl + nl), KR(kr, kr + nr);
    std::vector<uint8_t> DL(dl, dl + (size_t)32 * nl), DR(dr, dr + (size_t)32 * nr);
    StereoResult S = compute_stereo_matches(KL, DL, KR, DR, L->pyramid, Rr->pyramid, L->mvScaleFactor, L->mvInvScaleFactor, bf, fx);
    int m = 0;
    for (int i = 0; i < nl; i++) { uRight[i] = S.uRight[i]; depth[i] = S.depth[i]; best_sad[i] = S.best_sad[i]; if (S.uRight[i] >= 0) m++; }
    return m;
}
} // extern "C"

#include "local_ba.h"
extern "C" {
// Local BA on flat arrays: kfs[NK][22] (local first), preint[W][142], points[NP][3], edge_idx[NE][2] = (point, kf),
// edge_obs[NE][3] = u v invSigma2 (edges grouped by point). info6 = chi2_first chi2_final its_first its_second 0 0.
void ora_local_ba(const double* kfs, int nk, int n_local, int prev_kf, const double* preint, const double* points, int np_,
                  const int* edge_idx, const double* edge_obs, int ne, const double* gw, const double* cam16, const int* stop,
                  double* kfs_out, double* points_out, uint8_t* erase, double* info6) {
    BaProblem P;
    P.kfs.resize(nk); for (int i = 0; i < nk; i++) P.kfs[i] = ns_in(kfs + 22 * i);
    P.n_local = n_local; P.prev_kf = prev_kf;
    P.preint.resize(n_local); for (int i = 0; i < n_local; i++) P.preint[i] = pre_in(preint + 142 * i);
    P.points.resize(np_); for (int i = 0; i < np_; i++) P.points[i] = v3(points + 3 * i);
    P.edges.resize(ne); for (int k = 0; k < ne; k++) { P.edges[k].point = edge_idx[2 * k]; P.edges[k].kf = edge_idx[2 * k + 1]; P.edges[k].u = edge_obs[3 * k]; P.edges[k].v = edge_obs[3 * k + 1]; P.edges[k].inv_sigma2 = edge_obs[3 * k + 2]; }
    P.gw = v3(gw); P.cam = cam_in(cam16);
    BaResult R = local_ba_navstate(P, (const volatile int*)stop);
    for (int i = 0; i < n_local; i++) ns_out(kfs_out + 22 * i, R.kfs[i]);
    for (int i = 0; i < np_; i++) put3(points_out + 3 * i, R.points[i]);
    for (int k = 0; k < ne; k++) erase[k] = R.erase[k];
    info6[0] = R.chi2_after_first; info6[1] = R.chi2_final; info6[2] = R.its_first; info6[3] = R.its_second; info6[4] = 0; info6[5] = 0;
}
} // extern "C"

#include "bow.h"
extern "C" {
// DBoW2 transform over a flat vocabulary; word/weight/node per feature, plus the L1-normalised BowVector (sorted by word id).
int ora_bow_transform(int n_nodes, int L, const int32_t* child_start, const int32_t* child_ids, const uint8_t* vdesc, const int32_t* word_id,
                      const double* vweight, const uint8_t* desc, int n, int levelsup, int* word, double* weight, int* node,
                      int* bow_ids, double* bow_vals) {
    ora::Vocabulary V; V.n_nodes = n_nodes; V.L = L; V.child_start = child_start; V.child_ids = child_ids; V.desc = vdesc; V.word_id = word_id; V.weight = vweight;
    std::map<int, double> bow; std::map<int, std::vector<unsigned>> fv;
    ora::bow_transform(V, desc, n, levelsup, bow, fv, word, weight, node);
    int k = 0; for (auto& kv : bow) { bow_ids[k] = kv.first; bow_vals[k] = kv.second; k++; }
    return k;
}
// SearchByBoW with per-feature node ids (node < 0: feature absent from the FeatureVector)
int ora_search_by_bow(const uint8_t* kf_desc, const float* kf_angle, const int* kf_node, const uint8_t* kf_has_point, int nK,
                      const uint8_t* f_desc, const float* f_angle, const int* f_node, int nF, float nnratio, int check_ori, int* match) {
    std::map<int, std::vector<unsigned>> a, b;
    for (int i = 0; i < nK; i++) if (kf_node[i] >= 0) a[kf_node[i]].push_back((unsigned)i);
    for (int i = 0; i < nF; i++) if (f_node[i] >= 0) b[f_node[i]].push_back((unsigned)i);
    std::vector<int> m;
    const int n = ora::search_by_bow(a, kf_desc, kf_angle, kf_has_point, b, f_desc, f_angle, nF, nnratio, check_ori != 0, m);
    for (int i = 0; i < nF; i++) match[i] = m[i];
    return n;
}
} // extern "C"

#include "kf_matcher.h"
extern "C" {
int ora_search_for_triangulation(const ora::KeyPoint* k1, const uint8_t* d1, const uint8_t* hp1, const float* ur1, const int* node1, int n1,
                                 const ora::KeyPoint* k2, const uint8_t* d2, const uint8_t* hp2, const float* ur2, const int* node2, int n2,
                                 const float* F12, const float* Cw1, const float* pose12_2, const float* intr4, const float* sf2,
                                 const float* level_sigma2_2, int only_stereo, int check_ori, int* match12) {
    ora::KfView a, b;
    a.N = n1; a.kps = k1; a.desc = d1; a.has_point = hp1; a.uright = ur1; a.node = node1;
    b.N = n2; b.kps = k2; b.desc = d2; b.has_point = hp2; b.uright = ur2; b.node = node2;
    ora::PoseF T; for (int i = 0; i < 9; i++) T.Rcw[i] = pose12_2[i]; for (int i = 0; i < 3; i++) T.tcw[i] = pose12_2[9 + i];
    T.fx = intr4[0]; T.fy = intr4[1]; T.cx = intr4[2]; T.cy = intr4[3];
    std::vector<int> m;
    const int n = ora::search_for_triangulation(a, b, F12, Cw1, T, sf2, level_sigma2_2, only_stereo != 0, check_ori != 0, m);
    for (int i = 0; i < n1; i++) match12[i] = m[i];
    return n;
}
// pts_f[n][8] = Pw3 normal3 minDist maxDist; pts_valid[n]; intr5 = fx fy cx cy bf
int ora_fuse(const ora::KeyPoint* kps, const uint8_t* desc, const float* uright, int n, const float* bounds4, const float* pose12,
             const float* intr5, const float* sf, const float* inv_level_sigma2, int nlevels, float log_scale_factor, int npts,
             const float* pts_f, const uint8_t* pts_valid, const uint8_t* pts_desc, float th, int* best_idx) {
    ora::FrameGrid g; g.build(kps, desc, n, bounds4[0], bounds4[1], bounds4[2], bounds4[3]);
    ora::PoseF T; for (int i = 0; i < 9; i++) T.Rcw[i] = pose12[i]; for (int i = 0; i < 3; i++) T.tcw[i] = pose12[9 + i];
    T.fx = intr5[0]; T.fy = intr5[1]; T.cx = intr5[2]; T.cy = intr5[3];
    std::vector<ora::FusePoint> pts(npts);
    for (int i = 0; i < npts; i++) {
        pts[i].valid = pts_valid[i];
        for (int k = 0; k < 3; k++) { pts[i].Pw[k] = pts_f[8 * i + k]; pts[i].normal[k] = pts_f[8 * i + 3 + k]; }
        pts[i].min_dist = pts_f[8 * i + 6]; pts[i].max_dist = pts_f[8 * i + 7]; pts[i].desc = pts_desc + (size_t)32 * i;
    }
    std::vector<int> bi;
    const int nf = ora::fuse(g, uright, T, intr5[4], sf, inv_level_sigma2, nlevels, log_scale_factor, pts, th, bi);
    for (int i = 0; i < npts; i++) best_idx[i] = bi[i];
    return nf;
}
} // extern "C"

#include "local_ba_se3.h"
extern "C" {
// kfs[nk][7] = qx qy qz qw tx ty tz (g2o::SE3Quat of Tcw); edge_idx[ne][2] = (point, kf); edge_obs[ne][4] = u v ur invSigma2; intr5 = fx fy cx cy bf
int ora_local_ba_se3(const double* kfs, int nk, int n_local, const double* points, int np, const int* edge_idx, const double* edge_obs, int ne,
                     const double* intr5, const int* stop, double* kfs_out, double* points_out, uint8_t* erase, double* info) {
    ora::BaSe3Problem P; P.n_local = n_local; P.fx = intr5[0]; P.fy = intr5[1]; P.cx = intr5[2]; P.cy = intr5[3]; P.bf = intr5[4];
    P.kfs.resize(nk);
    for (int i = 0; i < nk; i++) { const double* k = kfs + 7 * i; P.kfs[i].r = ora::Quat{k[0], k[1], k[2], k[3]}; P.kfs[i].t = ora::V3{k[4], k[5], k[6]}; }
    P.points.resize(np); for (int i = 0; i < np; i++) P.points[i] = ora::V3{points[3 * i], points[3 * i + 1], points[3 * i + 2]};
    P.edges.resize(ne); for (int k = 0; k < ne; k++) P.edges[k] = ora::BaSe3Edge{edge_idx[2 * k], edge_idx[2 * k + 1], edge_obs[4 * k], edge_obs[4 * k + 1], edge_obs[4 * k + 2], edge_obs[4 * k + 3]};
    const ora::BaSe3Result R = ora::local_ba_se3(P, stop);
    for (int i = 0; i < n_local; i++) { double* k = kfs_out + 7 * i; k[0] = R.kfs[i].r.x; k[1] = R.kfs[i].r.y; k[2] = R.kfs[i].r.z; k[3] = R.kfs[i].r.w; k[4] = R.kfs[i].t.x; k[5] = R.kfs[i].t.y; k[6] = R.kfs[i].t.z; }
    for (int i = 0; i < np; i++) { points_out[3 * i] = R.points[i].x; points_out[3 * i + 1] = R.points[i].y; points_out[3 * i + 2] = R.points[i].z; }
    for (int k = 0; k < ne; k++) erase[k] = R.erase[k];
    info[0] = R.chi2_after_first; info[1] = R.chi2_final; info[2] = R.its_first; info[3] = R.its_second;
    return 0;
}
} // extern "C"

// ---- Frame::UndistortKeyPoints / ComputeImageBounds (undistort.cpp)
#include "undistort.h"
extern "C" {
void ora_undistort_points(const float* xy, int n, const float* K4, const float* dist5, float* xy_out) { ora::undistort_points(xy, n, K4, dist5, xy_out); }
void ora_image_bounds(int width, int height, const float* K4, const float* dist5, float* bounds4) { ora::image_bounds(width, height, K4, dist5, bounds4); }
}

// oracle/local_ba.cpp — TEST INFRASTRUCTURE ONLY (see local_ba.h).
#include "local_ba.h"
#include <limits>
#include <algorithm>
#include <array>
namespace ora {
namespace {
static void huber_(double e, double delta, double* rho) {
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; } else { const double sq = std::sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; }
}
static double fsq(double v) { return (double)(float)std::sqrt(v); }
struct Lin { double e[2]; double Jp[2][3]; double Jk[2][6]; };          // point block, (P, Phi) block of the key frame
// EdgeNavStatePVRPointXYZ::computeError / linearizeOplus
static V3 pc_of(const NavState& ns, const Camera& cam, V3 Pw) {
    const M3 Rcb = transpose(cam.Rbc);
    return Rcb * (transpose(ns.R.matrix()) * (Pw - ns.P)) - Rcb * cam.Pbc;
}
static void edge_error(const NavState& ns, const Camera& cam, V3 Pw, const BaEdge& ed, double* e) {
    const V3 Pc = pc_of(ns, cam, Pw);
    e[0] = ed.u - (Pc.x / Pc.z * cam.fx + cam.cx); e[1] = ed.v - (Pc.y / Pc.z * cam.fy + cam.cy);
}
static void edge_lin(const NavState& ns, const Camera& cam, V3 Pw, Lin& L) {
    const M3 Rcb = transpose(cam.Rbc), RwbT = transpose(ns.R.matrix());
    const V3 Pc = pc_of(ns, cam, Pw);
    const double x = Pc.x, y = Pc.y, z = Pc.z;
    const double Jpi[2][3] = {{cam.fx / z, 0, -x / z * cam.fx / z}, {0, cam.fy / z, -y / z * cam.fy / z}};
    const M3 RR = Rcb * RwbT;
    const V3 Paux = Rcb * (RwbT * (Pw - ns.P));
    const M3 HR = hat(Paux) * Rcb;
    for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) {
        double a = 0, b = 0, d = 0;
        for (int k = 0; k < 3; k++) { a += Jpi[r][k] * RR(k, c); b += Jpi[r][k] * Rcb(k, c); d += Jpi[r][k] * HR(k, c); }
        L.Jp[r][c] = -a;            // _jacobianOplusXi = -Jpi * Rcb * Rwb^T
        L.Jk[r][c] = b;             // JdPwb = -Jpi * (-Rcb)
        L.Jk[r][3 + c] = -d;        // JdRwb = -Jpi * (hat(Paux) * Rcb)
    }
}
} // namespace

BaResult local_ba_navstate(const BaProblem& P, const volatile int* stop) {
    const int W = P.n_local, NP = (int)P.points.size(), NE = (int)P.edges.size();
    const int np = 12 * W;                                   // pose unknowns: [PVR(9) | bias(3)] per local KF
    BaResult R; R.kfs.assign(P.kfs.begin(), P.kfs.begin() + W); R.points = P.points; R.erase.assign(NE, 0);
    auto terminate = [&]() { return stop && *stop; };
    if (terminate()) return R;
    std::vector<NavState> kf = P.kfs;                       // estimates (fixed ones never change)
    std::vector<V3> pt = P.points;
    std::vector<int> level(NE, 0);
    std::vector<double> err(2 * (size_t)NE, 0.0);           // g2o keeps _error per edge; only active edges are refreshed
    bool mono_kernel = true;
    const double d_mono = fsq(5.991), d_pvr = fsq(21.666), d_bias = fsq(16.812);
    // IMU factor information = cov^-1 (no inflation), bias factor information = I / accBiasRW2 / dt
    std::vector<Mat> info_pvr(W);
    for (int i = 0; i < W; i++) inverse(P.preint[i].cov, info_pvr[i]);
    auto pred = [&](int i) { return i == 0 ? P.prev_kf : i - 1; };       // predecessor key frame of local KF i
    std::vector<double> e_pvr(9 * (size_t)W), e_b(3 * (size_t)W);

    auto compute_active_errors = [&]() {
        for (int k = 0; k < NE; k++) if (level[k] == 0) edge_error(kf[P.edges[k].kf], P.cam, pt[P.edges[k].point], P.edges[k], &err[2 * k]);
        for (int i = 0; i < W; i++) {
            const int j = pred(i); if (j < 0) continue;
            edge_pvr_error(kf[j], kf[i], kf[j], P.preint[i], P.gw, &e_pvr[9 * i]);
            const V3 r = (kf[i].ba + kf[i].dba) - (kf[j].ba + kf[j].dba);
            e_b[3 * i] = r.x; e_b[3 * i + 1] = r.y; e_b[3 * i + 2] = r.z;
        }
    };
    auto chi_pvr = [&](int i) { double s = 0; for (int a = 0; a < 9; a++) { double t = 0; for (int b = 0; b < 9; b++) t += info_pvr[i](a, b) * e_pvr[9 * i + b]; s += e_pvr[9 * i + a] * t; } return s; };
    auto chi_b = [&](int i) { const double w = 1.0 / ImuNoise::accBiasRw2 / P.preint[i].dt; return w * (e_b[3 * i] * e_b[3 * i] + e_b[3 * i + 1] * e_b[3 * i + 1] + e_b[3 * i + 2] * e_b[3 * i + 2]); };
    auto chi_e = [&](int k) { return P.edges[k].inv_sigma2 * (err[2 * k] * err[2 * k] + err[2 * k + 1] * err[2 * k + 1]); };
    auto robust_chi2 = [&]() {
        double c = 0, r[2];
        for (int k = 0; k < NE; k++) if (level[k] == 0) { if (mono_kernel) { huber_(chi_e(k), d_mono, r); c += r[0]; } else c += chi_e(k); }
        for (int i = 0; i < W; i++) if (pred(i) >= 0) { huber_(chi_pvr(i), d_pvr, r); c += r[0]; huber_(chi_b(i), d_bias, r); c += r[0]; }
        return c;
    };
    Mat Hpp; std::vector<double> bp, bl; std::vector<M3> Hll; std::vector<Lin> lin(NE); std::vector<double> wgt(NE);
    auto build_system = [&]() {
        Hpp = Mat(np, np); bp.assign(np, 0.0); bl.assign(3 * (size_t)NP, 0.0); Hll.assign(NP, M3());
        for (int k = 0; k < NE; k++) {
            if (level[k] != 0) continue;
            const BaEdge& ed = P.edges[k];
            edge_lin(kf[ed.kf], P.cam, pt[ed.point], lin[k]);
            double w = 1.0, r[2]; if (mono_kernel) { huber_(chi_e(k), d_mono, r); w = r[1]; }
            w *= ed.inv_sigma2; wgt[k] = w;
            const Lin& L = lin[k]; const double* e = &err[2 * k];
            for (int a = 0; a < 3; a++) { bl[3 * ed.point + a] -= w * (L.Jp[0][a] * e[0] + L.Jp[1][a] * e[1]); for (int b = 0; b < 3; b++) Hll[ed.point](a, b) += w * (L.Jp[0][a] * L.Jp[0][b] + L.Jp[1][a] * L.Jp[1][b]); }
            if (ed.kf < W) {
                const int base = 12 * ed.kf; const int loc[6] = {0, 1, 2, 6, 7, 8};
                for (int a = 0; a < 6; a++) { bp[base + loc[a]] -= w * (L.Jk[0][a] * e[0] + L.Jk[1][a] * e[1]); for (int b = 0; b < 6; b++) Hpp(base + loc[a], base + loc[b]) += w * (L.Jk[0][a] * L.Jk[0][b] + L.Jk[1][a] * L.Jk[1][b]); }
            }
        }
        for (int i = 0; i < W; i++) {
            const int j = pred(i); if (j < 0) continue;
            // IMU factor: vertices (PVR_j, PVR_i, Bias_j); columns map to x when the vertex is a local key frame
            Mat Ji, Jj, Jb; edge_pvr_jacobians(kf[j], kf[i], kf[j], P.preint[i], P.gw, &e_pvr[9 * i], Ji, Jj, Jb);
            double r[2]; huber_(chi_pvr(i), d_pvr, r); const double w = r[1];
            std::vector<int> map(21, -1);
            for (int c = 0; c < 9; c++) { if (j < W) map[c] = 12 * j + c; map[9 + c] = 12 * i + c; }
            for (int c = 0; c < 3; c++) if (j < W) map[18 + c] = 12 * j + 9 + c;
            Mat J(9, 21);
            for (int a = 0; a < 9; a++) { for (int c = 0; c < 9; c++) { J(a, c) = Ji(a, c); J(a, 9 + c) = Jj(a, c); } for (int c = 0; c < 3; c++) J(a, 18 + c) = Jb(a, c); }
            Mat OJ = info_pvr[i] * J;
            for (int a = 0; a < 21; a++) { if (map[a] < 0) continue;
                double s = 0; for (int q = 0; q < 9; q++) s += OJ(q, a) * e_pvr[9 * i + q];
                bp[map[a]] -= w * s;
                for (int b = 0; b < 21; b++) { if (map[b] < 0) continue; double t = 0; for (int q = 0; q < 9; q++) t += J(q, a) * OJ(q, b); Hpp(map[a], map[b]) += w * t; } }
            // bias factor: J_j = -I, J_i = +I
            huber_(chi_b(i), d_bias, r); const double wb = r[1] / ImuNoise::accBiasRw2 / P.preint[i].dt;
            for (int c = 0; c < 3; c++) {
                const int ic = 12 * i + 9 + c, jc = j < W ? 12 * j + 9 + c : -1;
                Hpp(ic, ic) += wb; bp[ic] -= wb * e_b[3 * i + c];
                if (jc >= 0) { Hpp(jc, jc) += wb; Hpp(ic, jc) -= wb; Hpp(jc, ic) -= wb; bp[jc] += wb * e_b[3 * i + c]; }
            }
        }
    };
    std::vector<double> xp, xl;
    // BlockSolver::solve with Schur complement at damping lambda; false if the reduced system is not positive definite
    auto solve = [&](double lambda) -> bool {
        Mat S = Hpp; for (int i = 0; i < np; i++) S(i, i) += lambda;
        std::vector<double> bs = bp;
        std::vector<M3> Dinv(NP);
        for (int p = 0; p < NP; p++) { Mat D(3, 3), Di; for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) D(a, b) = Hll[p](a, b) + (a == b ? lambda : 0.0); inverse(D, Di); for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Dinv[p](a, b) = Di(a, b); }
        // edges are grouped by point
        int k0 = 0;
        const int loc[6] = {0, 1, 2, 6, 7, 8};
        std::vector<std::array<double, 18>> Wb(NE);              // Hpl block of edge k: 6 x 3
        for (int k = 0; k < NE; k++) if (level[k] == 0 && P.edges[k].kf < W) { const Lin& L = lin[k]; for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++) Wb[k][a * 3 + b] = wgt[k] * (L.Jk[0][a] * L.Jp[0][b] + L.Jk[1][a] * L.Jp[1][b]); }
        while (k0 < NE) {
            int k1 = k0; const int p = P.edges[k0].point; while (k1 < NE && P.edges[k1].point == p) k1++;
            const V3 db = Dinv[p] * V3{bl[3 * p], bl[3 * p + 1], bl[3 * p + 2]};
            for (int a = k0; a < k1; a++) { if (level[a] != 0 || P.edges[a].kf >= W) continue;
                double BD[18]; for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int q = 0; q < 3; q++) s += Wb[a][r * 3 + q] * Dinv[p](q, c); BD[r * 3 + c] = s; }
                const int ba = 12 * P.edges[a].kf;
                for (int r = 0; r < 6; r++) bs[ba + loc[r]] -= Wb[a][r * 3] * db.x + Wb[a][r * 3 + 1] * db.y + Wb[a][r * 3 + 2] * db.z;
                for (int b = k0; b < k1; b++) { if (level[b] != 0 || P.edges[b].kf >= W) continue;
                    const int bb = 12 * P.edges[b].kf;
                    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) { double s = 0; for (int q = 0; q < 3; q++) s += BD[r * 3 + q] * Wb[b][c * 3 + q]; S(ba + loc[r], bb + loc[c]) -= s; } } }
            k0 = k1;
        }
        if (!cholesky_solve(S, bs, xp)) return false;
        xl.assign(3 * (size_t)NP, 0.0);
        std::vector<double> cl = bl;
        for (int k = 0; k < NE; k++) if (level[k] == 0 && P.edges[k].kf < W) { const int ba = 12 * P.edges[k].kf, p = P.edges[k].point; for (int c = 0; c < 3; c++) { double s = 0; for (int r = 0; r < 6; r++) s += Wb[k][r * 3 + c] * xp[ba + loc[r]]; cl[3 * p + c] -= s; } }
        for (int p = 0; p < NP; p++) { const V3 v = Dinv[p] * V3{cl[3 * p], cl[3 * p + 1], cl[3 * p + 2]}; xl[3 * p] = v.x; xl[3 * p + 1] = v.y; xl[3 * p + 2] = v.z; }
        return true;
    };
    double lambda = 0, ni = 2;
    auto optimize = [&](int iterations, int& its_done) -> double {
        double currentChi = 0; int nBad = 0;
        // points without any active edge and key frames are all "active"; g2o only indexes vertices with active edges,
        // a point whose edges are all at level 1 keeps its estimate (Hll = 0 would be singular): freeze it
        for (int it = 0; it < iterations && !terminate(); it++) {
            compute_active_errors();
            currentChi = robust_chi2(); const double iniChi = currentChi;
            build_system();
            if (it == 0) { double mx = 0; for (int i = 0; i < np; i++) mx = std::max(std::fabs(Hpp(i, i)), mx); for (int p = 0; p < NP; p++) for (int a = 0; a < 3; a++) mx = std::max(std::fabs(Hll[p](a, a)), mx); lambda = 1e-5 * mx; ni = 2; nBad = 0; }
            double rho = 0; int qmax = 0;
            do {
                const std::vector<NavState> bk = kf; const std::vector<V3> bp_ = pt;
                const bool ok2 = solve(lambda);
                if (ok2) {
                    for (int i = 0; i < W; i++) { kf[i].inc_small_pvr(&xp[12 * i]); kf[i].inc_small_bias(&xp[12 * i + 9]); }
                    for (int p = 0; p < NP; p++) { bool act = false; (void)act; pt[p] = pt[p] + V3{xl[3 * p], xl[3 * p + 1], xl[3 * p + 2]}; }
                }
                compute_active_errors();
                double tempChi = robust_chi2();
                if (!ok2) tempChi = std::numeric_limits<double>::max();
                double scale = 0;
                if (ok2) { for (int j = 0; j < np; j++) scale += xp[j] * (lambda * xp[j] + bp[j]); for (size_t j = 0; j < xl.size(); j++) scale += xl[j] * (lambda * xl[j] + bl[j]); }
                scale += 1e-3;
                rho = (currentChi - tempChi) / scale;
                if (rho > 0 && std::isfinite(tempChi)) { double alpha = 1. - std::pow((2 * rho - 1), 3); alpha = std::min(alpha, 2. / 3.); lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi; }
                else { lambda *= ni; ni *= 2; kf = bk; pt = bp_; }
                qmax++;
            } while (rho < 0 && qmax < 10 && !terminate());
            its_done++; R.trace.push_back(currentChi);
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
            if (nBad >= 3) break;
        }
        return currentChi;
    };
    R.chi2_after_first = optimize(5, R.its_first);
    if (!terminate()) {
        for (int k = 0; k < NE; k++) {                       // :2037-2051
            const V3 Pc = pc_of(kf[P.edges[k].kf], P.cam, pt[P.edges[k].point]);
            if (chi_e(k) > 5.991 || !(Pc.z > 0.0)) level[k] = 1;
        }
        mono_kernel = false;
        // a point left without active edges has Hll = 0: g2o drops it from the active set; give it an identity block
        R.chi2_final = optimize(10, R.its_second);
    }
    for (int k = 0; k < NE; k++) {                           // :2105-2118 (stale _error on excluded edges, fresh depth)
        const V3 Pc = pc_of(kf[P.edges[k].kf], P.cam, pt[P.edges[k].point]);
        R.erase[k] = (chi_e(k) > 5.991 || !(Pc.z > 0.0)) ? 1 : 0;
    }
    for (int i = 0; i < W; i++) R.kfs[i] = kf[i];
    R.points = pt;
    return R;
}
}

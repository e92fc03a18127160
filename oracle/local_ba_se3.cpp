// oracle/local_ba_se3.cpp — TEST INFRASTRUCTURE ONLY (see local_ba_se3.h).
#include "local_ba_se3.h"
#include <limits>
#include <algorithm>
#include <array>
namespace ora {
namespace {
void huber_(double e, double delta, double* rho) {
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; } else { const double sq = std::sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; }
}
double fsq(double v) { return (double)(float)std::sqrt(v); }
void norm_rot(Se3Pose& s) { if (s.r.w < 0) { s.r.x = -s.r.x; s.r.y = -s.r.y; s.r.z = -s.r.z; s.r.w = -s.r.w; } s.r = normalized(s.r); }
V3 map_pt(const Se3Pose& s, V3 p) { return rotate(s.r, p) + s.t; }
Se3Pose se3_mul(const Se3Pose& a, const Se3Pose& b) { Se3Pose o = a; o.t = o.t + rotate(a.r, b.t); o.r = a.r * b.r; norm_rot(o); return o; }
Se3Pose se3_exp(const double* u) {                                  // se3quat.h:223-257
    const V3 omega{u[0], u[1], u[2]}, ups{u[3], u[4], u[5]};
    const double theta = norm(omega);
    const M3 Om = hat(omega);
    M3 R, Vm;
    if (theta < 0.00001) { R = M3::identity() + Om + Om * Om; Vm = R; }
    else {
        const M3 Om2 = Om * Om;
        R = M3::identity() + Om * (std::sin(theta) / theta) + Om2 * ((1 - std::cos(theta)) / (theta * theta));
        Vm = M3::identity() + Om * ((1 - std::cos(theta)) / (theta * theta)) + Om2 * ((theta - std::sin(theta)) / std::pow(theta, 3));
    }
    Se3Pose o; o.r = quat_from_matrix(R); o.t = Vm * ups; norm_rot(o); return o;
}
struct Lin { double e[3]; double Jp[3][3]; double Jk[3][6]; };
// computeError of both edge types (types_six_dof_expmap.cpp:85-101,141-157)
void edge_error(const BaSe3Problem& P, const Se3Pose& T, V3 Xw, const BaSe3Edge& ed, double* e) {
    const V3 p = map_pt(T, Xw);
    if (ed.ur < 0) { e[0] = ed.u - (p.x / p.z * P.fx + P.cx); e[1] = ed.v - (p.y / p.z * P.fy + P.cy); e[2] = 0; }
    else {
        const float invz = (float)(1.0 / p.z);                          // double division, one rounding to float (types_six_dof_expmap.cpp:151)
        const double r0 = p.x * invz * P.fx + P.cx, r1 = p.y * invz * P.fy + P.cy, r2 = r0 - P.bf * invz;
        e[0] = ed.u - r0; e[1] = ed.v - r1; e[2] = ed.ur - r2;
    }
}
void edge_lin(const BaSe3Problem& P, const Se3Pose& T, V3 Xw, bool stereo, Lin& L) {
    const V3 p = map_pt(T, Xw);
    const M3 R = SO3(T.r).matrix();
    const double x = p.x, y = p.y, z = p.z, z_2 = z * z, fx = P.fx, fy = P.fy, bf = P.bf;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) L.Jp[r][c] = 0;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 6; c++) L.Jk[r][c] = 0;
    if (!stereo) {
        const double tmp[2][3] = {{fx, 0, -x / z * fx}, {0, fy, -y / z * fy}};
        for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += (-1. / z * tmp[r][k]) * R(k, c); L.Jp[r][c] = s; }
    } else {
        for (int c = 0; c < 3; c++) {
            L.Jp[0][c] = -fx * R(0, c) / z + fx * x * R(2, c) / z_2;
            L.Jp[1][c] = -fy * R(1, c) / z + fy * y * R(2, c) / z_2;
            L.Jp[2][c] = L.Jp[0][c] - bf * R(2, c) / z_2;
        }
    }
    L.Jk[0][0] = x * y / z_2 * fx; L.Jk[0][1] = -(1 + (x * x / z_2)) * fx; L.Jk[0][2] = y / z * fx; L.Jk[0][3] = -1. / z * fx; L.Jk[0][4] = 0; L.Jk[0][5] = x / z_2 * fx;
    L.Jk[1][0] = (1 + y * y / z_2) * fy; L.Jk[1][1] = -x * y / z_2 * fy; L.Jk[1][2] = -x / z * fy; L.Jk[1][3] = 0; L.Jk[1][4] = -1. / z * fy; L.Jk[1][5] = y / z_2 * fy;
    if (stereo) {
        L.Jk[2][0] = L.Jk[0][0] - bf * y / z_2; L.Jk[2][1] = L.Jk[0][1] + bf * x / z_2; L.Jk[2][2] = L.Jk[0][2]; L.Jk[2][3] = L.Jk[0][3]; L.Jk[2][4] = 0;
        L.Jk[2][5] = L.Jk[0][5] - bf / z_2;
    }
}
} // namespace

BaSe3Result local_ba_se3(const BaSe3Problem& P, const volatile int* stop) {
    const int W = P.n_local, NP = (int)P.points.size(), NE = (int)P.edges.size(), np = 6 * W;
    BaSe3Result R; R.kfs.assign(P.kfs.begin(), P.kfs.begin() + W); R.points = P.points; R.erase.assign(NE, 0);
    auto terminate = [&]() { return stop && *stop; };
    if (terminate()) return R;
    std::vector<Se3Pose> kf = P.kfs; std::vector<V3> pt = P.points;
    std::vector<int> level(NE, 0); std::vector<double> err(3 * (size_t)NE, 0.0);
    bool kernels = true;
    const double d_mono = fsq(5.991), d_stereo = fsq(7.815);
    auto chi_e = [&](int k) { return P.edges[k].inv_sigma2 * (err[3 * k] * err[3 * k] + err[3 * k + 1] * err[3 * k + 1] + err[3 * k + 2] * err[3 * k + 2]); };
    auto compute_active_errors = [&]() { for (int k = 0; k < NE; k++) if (level[k] == 0) edge_error(P, kf[P.edges[k].kf], pt[P.edges[k].point], P.edges[k], &err[3 * k]); };
    auto robust_chi2 = [&]() {
        double c = 0, r[2];
        for (int k = 0; k < NE; k++) if (level[k] == 0) { if (kernels) { huber_(chi_e(k), P.edges[k].ur < 0 ? d_mono : d_stereo, r); c += r[0]; } else c += chi_e(k); }
        return c;
    };
    Mat Hpp; std::vector<double> bp, bl; std::vector<M3> Hll; std::vector<Lin> lin(NE); std::vector<double> wgt(NE);
    auto build_system = [&]() {
        Hpp = Mat(np, np); bp.assign(np, 0.0); bl.assign(3 * (size_t)NP, 0.0); Hll.assign(NP, M3());
        for (int k = 0; k < NE; k++) {
            if (level[k] != 0) continue;
            const BaSe3Edge& ed = P.edges[k];
            edge_lin(P, kf[ed.kf], pt[ed.point], ed.ur >= 0, lin[k]);
            double w = 1.0, r[2]; if (kernels) { huber_(chi_e(k), ed.ur < 0 ? d_mono : d_stereo, r); w = r[1]; }
            w *= ed.inv_sigma2; wgt[k] = w;
            const Lin& L = lin[k]; const double* e = &err[3 * k];
            for (int a = 0; a < 3; a++) {
                double s = 0; for (int q = 0; q < 3; q++) s += L.Jp[q][a] * e[q];
                bl[3 * ed.point + a] -= w * s;
                for (int b = 0; b < 3; b++) { double t = 0; for (int q = 0; q < 3; q++) t += L.Jp[q][a] * L.Jp[q][b]; Hll[ed.point](a, b) += w * t; }
            }
            if (ed.kf < W) {
                const int base = 6 * ed.kf;
                for (int a = 0; a < 6; a++) {
                    double s = 0; for (int q = 0; q < 3; q++) s += L.Jk[q][a] * e[q];
                    bp[base + a] -= w * s;
                    for (int b = 0; b < 6; b++) { double t = 0; for (int q = 0; q < 3; q++) t += L.Jk[q][a] * L.Jk[q][b]; Hpp(base + a, base + b) += w * t; }
                }
            }
        }
    };
    std::vector<double> xp, xl;
    auto solve = [&](double lambda) -> bool {
        Mat S = Hpp; for (int i = 0; i < np; i++) S(i, i) += lambda;
        std::vector<double> bs = bp; std::vector<M3> Dinv(NP);
        for (int p = 0; p < NP; p++) { Mat D(3, 3), Di; for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) D(a, b) = Hll[p](a, b) + (a == b ? lambda : 0.0); inverse(D, Di); for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Dinv[p](a, b) = Di(a, b); }
        std::vector<std::array<double, 18>> Wb(NE);
        for (int k = 0; k < NE; k++) if (level[k] == 0 && P.edges[k].kf < W) { const Lin& L = lin[k]; for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++) { double t = 0; for (int q = 0; q < 3; q++) t += L.Jk[q][a] * L.Jp[q][b]; Wb[k][a * 3 + b] = wgt[k] * t; } }
        int k0 = 0;
        while (k0 < NE) {
            int k1 = k0; const int p = P.edges[k0].point; while (k1 < NE && P.edges[k1].point == p) k1++;
            const V3 db = Dinv[p] * V3{bl[3 * p], bl[3 * p + 1], bl[3 * p + 2]};
            for (int a = k0; a < k1; a++) { if (level[a] != 0 || P.edges[a].kf >= W) continue;
                double BD[18]; for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int q = 0; q < 3; q++) s += Wb[a][r * 3 + q] * Dinv[p](q, c); BD[r * 3 + c] = s; }
                const int ba = 6 * P.edges[a].kf;
                for (int r = 0; r < 6; r++) bs[ba + r] -= Wb[a][r * 3] * db.x + Wb[a][r * 3 + 1] * db.y + Wb[a][r * 3 + 2] * db.z;
                for (int b = k0; b < k1; b++) { if (level[b] != 0 || P.edges[b].kf >= W) continue;
                    const int bb = 6 * P.edges[b].kf;
                    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) { double s = 0; for (int q = 0; q < 3; q++) s += BD[r * 3 + q] * Wb[b][c * 3 + q]; S(ba + r, bb + c) -= s; } } }
            k0 = k1;
        }
        if (!cholesky_solve(S, bs, xp)) return false;
        xl.assign(3 * (size_t)NP, 0.0);
        std::vector<double> cl = bl;
        for (int k = 0; k < NE; k++) if (level[k] == 0 && P.edges[k].kf < W) { const int ba = 6 * P.edges[k].kf, p = P.edges[k].point; for (int c = 0; c < 3; c++) { double s = 0; for (int r = 0; r < 6; r++) s += Wb[k][r * 3 + c] * xp[ba + r]; cl[3 * p + c] -= s; } }
        for (int p = 0; p < NP; p++) { const V3 v = Dinv[p] * V3{cl[3 * p], cl[3 * p + 1], cl[3 * p + 2]}; xl[3 * p] = v.x; xl[3 * p + 1] = v.y; xl[3 * p + 2] = v.z; }
        return true;
    };
    double lambda = 0, ni = 2;
    auto optimize = [&](int iterations, int& its_done) -> double {
        double currentChi = 0; int nBad = 0;
        for (int it = 0; it < iterations && !terminate(); it++) {
            compute_active_errors();
            currentChi = robust_chi2(); const double iniChi = currentChi;
            build_system();
            if (it == 0) { double mx = 0; for (int i = 0; i < np; i++) mx = std::max(std::fabs(Hpp(i, i)), mx); for (int p = 0; p < NP; p++) for (int a = 0; a < 3; a++) mx = std::max(std::fabs(Hll[p](a, a)), mx); lambda = 1e-5 * mx; ni = 2; nBad = 0; }
            double rho = 0; int qmax = 0;
            do {
                const std::vector<Se3Pose> bk = kf; const std::vector<V3> bpt = pt;
                const bool ok2 = solve(lambda);
                if (ok2) {
                    for (int i = 0; i < W; i++) kf[i] = se3_mul(se3_exp(&xp[6 * i]), kf[i]);           // VertexSE3Expmap::oplusImpl
                    for (int p = 0; p < NP; p++) pt[p] = pt[p] + V3{xl[3 * p], xl[3 * p + 1], xl[3 * p + 2]};
                }
                compute_active_errors();
                double tempChi = robust_chi2();
                if (!ok2) tempChi = std::numeric_limits<double>::max();
                double scale = 0;
                if (ok2) { for (int j = 0; j < np; j++) scale += xp[j] * (lambda * xp[j] + bp[j]); for (size_t j = 0; j < xl.size(); j++) scale += xl[j] * (lambda * xl[j] + bl[j]); }
                scale += 1e-3;
                rho = (currentChi - tempChi) / scale;
                if (rho > 0 && std::isfinite(tempChi)) { double alpha = 1. - std::pow((2 * rho - 1), 3); alpha = std::min(alpha, 2. / 3.); lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi; }
                else { lambda *= ni; ni *= 2; kf = bk; pt = bpt; }
                qmax++;
            } while (rho < 0 && qmax < 10 && !terminate());
            its_done++;
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
            if (nBad >= 3) break;
        }
        return currentChi;
    };
    auto bad_edge = [&](int k) {
        const V3 pc = map_pt(kf[P.edges[k].kf], pt[P.edges[k].point]);
        return chi_e(k) > (P.edges[k].ur < 0 ? 5.991 : 7.815) || !(pc.z > 0.0);
    };
    R.chi2_after_first = optimize(5, R.its_first);
    if (!terminate()) {
        for (int k = 0; k < NE; k++) if (bad_edge(k)) level[k] = 1;       // :4170-4200; kernels dropped on every edge
        kernels = false;
        R.chi2_final = optimize(10, R.its_second);
    }
    for (int k = 0; k < NE; k++) R.erase[k] = bad_edge(k) ? 1 : 0;         // :4207-4235 (stale _error on excluded edges, fresh depth)
    for (int i = 0; i < W; i++) R.kfs[i] = kf[i];
    R.points = pt;
    return R;
}
}

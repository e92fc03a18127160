// oracle/vio.cpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md, vio.h for the parity statement).
#include "vio.h"
#include <algorithm>
#include <limits>
#include <functional>

namespace ora {

// ------------------------------------------------------------------------------------------------
// IMU pre-integration — reference src/IMU/IMUPreintegrator.cpp:61-153
// ------------------------------------------------------------------------------------------------
void Preint::reset() {
    dP = V3(); dV = V3(); dR = M3::identity();
    JPg = M3(); JPa = M3(); JVg = M3(); JVa = M3(); JRg = M3();
    cov = Mat(9, 9); dt = 0;
}

static M3 normalize_rotation(const M3& R) {           // normalizeRotationM (IMUPreintegrator.h)
    Quat q = quat_from_matrix(R);
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    return to_matrix(normalized(q));
}

void Preint::update(V3 omega, V3 acc, double d) {
    const double dt2 = d * d;
    const M3 dRk = SO3::exp(omega * d).matrix();
    const M3 Jr = jacobian_r(omega * d);
    const M3 I3 = M3::identity();
    // covariance propagation (:94-107)
    Mat A = Mat::identity(9);
    A.set_block(6, 6, transpose(dRk));
    A.set_block(3, 6, dR * hat(acc), -d);
    A.set_block(0, 6, dR * hat(acc), -0.5 * dt2);
    A.set_block(0, 3, I3, d);
    Mat Bg(9, 3); Bg.set_block(6, 0, Jr, d);
    Mat Ca(9, 3); Ca.set_block(3, 0, dR, d); Ca.set_block(0, 0, dR, 0.5 * dt2);
    Mat Sg = Mat::identity(3), Sa = Mat::identity(3);
    for (int i = 0; i < 3; i++) { Sg(i, i) = ImuNoise::gyrMeasCov; Sa(i, i) = ImuNoise::accMeasCov; }
    cov = A * cov * transpose(A) + Bg * Sg * transpose(Bg) + Ca * Sa * transpose(Ca);
    // bias Jacobians (:111-115): P first, then V, then R
    JPa = JPa + JVa * d - dR * (0.5 * dt2);
    JPg = JPg + JVg * d - (dR * hat(acc) * JRg) * (0.5 * dt2);
    JVa = JVa - dR * d;
    JVg = JVg - (dR * hat(acc) * JRg) * d;
    JRg = transpose(dRk) * JRg - Jr * d;
    // deltas (:119-121)
    dP = dP + dV * d + (dR * acc) * (0.5 * dt2);
    dV = dV + (dR * acc) * d;
    dR = normalize_rotation(dR * dRk);
    dt += d;
}

void preintegrate(const ImuSample* s, int n, V3 bg, V3 ba, double t_last, double t_cur, Preint& out) {
    out.reset();
    if (n <= 0) return;
    auto gyr = [&](int i) { return V3{s[i].g[0], s[i].g[1], s[i].g[2]} - bg; };
    auto acc = [&](int i) { return V3{s[i].a[0], s[i].a[1], s[i].a[2]} - ba; };
    out.update(gyr(0), acc(0), s[0].t - t_last);
    for (int i = 0; i < n; i++) {
        const double nextt = (i == n - 1) ? t_cur : s[i + 1].t;
        out.update(gyr(i), acc(i), nextt - s[i].t);
    }
}

// ------------------------------------------------------------------------------------------------
// NavState — reference src/IMU/NavState.cpp:71-140, src/Converter.cc:27-49
// ------------------------------------------------------------------------------------------------
void NavState::inc_small_pvr(const double* u) {
    const M3 Rm = R.matrix();
    P = P + Rm * V3{u[0], u[1], u[2]};
    V = V + V3{u[3], u[4], u[5]};
    R = R * SO3::exp(V3{u[6], u[7], u[8]});
}
void NavState::inc_small_bias(const double* u) { dba = dba + V3{u[0], u[1], u[2]}; }

void update_ns(NavState& ns, const Preint& p, V3 gw) {
    const V3 Pwbpre = ns.P, Vwbpre = ns.V; const M3 Rwbpre = ns.R.matrix();
    const double dt = p.dt;
    const M3 Rwb = Rwbpre * p.dR;
    const V3 Pwb = Pwbpre + Vwbpre * dt + ((gw * 0.5) * dt) * dt + Rwbpre * p.dP;
    const V3 Vwb = Vwbpre + gw * dt + Rwbpre * p.dV;
    ns.P = Pwb; ns.V = Vwb; ns.R = SO3(Rwb);
}

NavState predict_navstate(const NavState& last, const Preint& p, V3 gw) {
    NavState ns = last;
    ns.bg = last.bg + last.dbg; ns.ba = last.ba + last.dba; ns.dbg = V3(); ns.dba = V3();
    update_ns(ns, p, gw);
    return ns;
}

void pose_from_navstate_f32(const NavState& ns, const Camera& cam, float* pose12) {
    const M3 Rd = ns.R.matrix();
    float Rwb[3][3], Rbc[3][3], Pbc[3] = {(float)cam.Pbc.x, (float)cam.Pbc.y, (float)cam.Pbc.z};
    const float Pwb[3] = {(float)ns.P.x, (float)ns.P.y, (float)ns.P.z};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rwb[i][j] = (float)Rd(i, j); Rbc[i][j] = (float)cam.Rbc(i, j); }
    float Rwc[3][3], Pwc[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Rwc[i][j] = Rwb[i][0] * Rbc[0][j] + Rwb[i][1] * Rbc[1][j] + Rwb[i][2] * Rbc[2][j];   // (Rwb*Rbc)
        const float t = Rwb[i][0] * Pbc[0] + Rwb[i][1] * Pbc[1] + Rwb[i][2] * Pbc[2];
        Pwc[i] = (float)((double)t + (double)Pwb[i]);                                                                     // Rwb*Pbc + Pwb
    }
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) pose12[3 * i + j] = Rwc[j][i];                                                        // .t()
        const float t = Rwc[0][i] * Pwc[0] + Rwc[1][i] * Pwc[1] + Rwc[2][i] * Pwc[2];
        pose12[9 + i] = -t;                                                                                               // -Rcw*Pwc
    }
}

// ------------------------------------------------------------------------------------------------
// Edges — reference src/IMU/g2otypes.{h,cpp}
// ------------------------------------------------------------------------------------------------
// EdgeNavStatePVR::computeError, g2otypes.cpp:8-75
void edge_pvr_error(const NavState& ni, const NavState& nj, const NavState& nb, const Preint& M, V3 gw, double* e) {
    const double dT = M.dt, dT2 = dT * dT;
    const SO3 RiT = ni.R.inverse();
    const V3 rP = RiT * (nj.P - ni.P - ni.V * dT - gw * (0.5 * dT2)) - (M.dP + M.JPg * nb.dbg + M.JPa * nb.dba);
    const V3 rV = RiT * (nj.V - ni.V - gw * dT) - (M.dV + M.JVg * nb.dbg + M.JVa * nb.dba);
    const SO3 dRij(M.dR);
    const SO3 dR_dbg = SO3::exp(M.JRg * nb.dbg);
    const SO3 rR = (dRij * dR_dbg).inverse() * RiT * nj.R;
    const V3 rPhi = rR.log();
    e[0] = rP.x; e[1] = rP.y; e[2] = rP.z; e[3] = rV.x; e[4] = rV.y; e[5] = rV.z; e[6] = rPhi.x; e[7] = rPhi.y; e[8] = rPhi.z;
}
// EdgeNavStatePVR::linearizeOplus, g2otypes.cpp:77-229 (NOT_UPDATE_GYRO_BIAS: bias block is 9x3)
void edge_pvr_jacobians(const NavState& ni, const NavState& nj, const NavState& nb, const Preint& M, V3 gw,
                        const double* e, Mat& Ji, Mat& Jj, Mat& Jb) {
    const double dT = M.dt, dT2 = dT * dT;
    const M3 Ri = ni.R.matrix(), Rj = nj.R.matrix(), RiT = transpose(Ri), RjT = transpose(Rj), I3 = M3::identity();
    const V3 rPhi{e[6], e[7], e[8]};
    const M3 JrInv = jacobian_r_inv(rPhi);
    Ji = Mat(9, 9); Jj = Mat(9, 9); Jb = Mat(9, 3);
    Ji.set_block(0, 0, I3, -1); Ji.set_block(0, 3, RiT, -dT);
    Ji.set_block(0, 6, hat(RiT * (nj.P - ni.P - ni.V * dT - gw * (0.5 * dT2))));
    Ji.set_block(3, 3, RiT, -1);
    Ji.set_block(3, 6, hat(RiT * (nj.V - ni.V - gw * dT)));
    Ji.set_block(6, 6, JrInv * RjT * Ri, -1);
    Jj.set_block(0, 0, RiT * Rj); Jj.set_block(3, 3, RiT); Jj.set_block(6, 6, JrInv);
    Jb.set_block(0, 0, M.JPa, -1); Jb.set_block(3, 0, M.JVa, -1);
}
// EdgeNavStatePVRPointXYZOnlyPose, g2otypes.h:205-281 / g2otypes.cpp:356-407
static V3 proj_pc(const NavState& ns, const Camera& cam, V3 Pw) {
    const M3 Rcb = transpose(cam.Rbc);
    return Rcb * (transpose(ns.R.matrix()) * (Pw - ns.P)) - Rcb * cam.Pbc;
}
void edge_proj_error(const NavState& ns, const Camera& cam, const Observation& o, double* e) {
    const V3 Pc = proj_pc(ns, cam, o.Pw);
    e[0] = o.u - (Pc.x / Pc.z * cam.fx + cam.cx);
    e[1] = o.v - (Pc.y / Pc.z * cam.fy + cam.cy);
}
void edge_proj_jacobian(const NavState& ns, const Camera& cam, const Observation& o, Mat& J) {
    const M3 Rcb = transpose(cam.Rbc);
    const V3 Pc = proj_pc(ns, cam, o.Pw);
    const double x = Pc.x, y = Pc.y, z = Pc.z;
    double Jpi[2][3] = {{cam.fx / z, 0, -x / z * cam.fx / z}, {0, cam.fy / z, -y / z * cam.fy / z}};
    const V3 Paux = Rcb * (transpose(ns.R.matrix()) * (o.Pw - ns.P));
    const M3 HR = hat(Paux) * Rcb;
    J = Mat(2, 9);
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++) {
            double a = 0, b = 0;
            for (int k = 0; k < 3; k++) { a += Jpi[r][k] * Rcb(k, c); b += Jpi[r][k] * HR(k, c); }
            J(r, c) = a;           // JdPwb = -Jpi * (-Rcb)
            J(r, 6 + c) = -b;      // JdRwb = -Jpi * (hat(Paux) * Rcb)
        }
}
// EdgeNavStatePriorPVRBias (12-D), g2otypes.cpp:409-515
void edge_prior_error(const NavState& pvr, const NavState& bias, const NavState& prior, double* e) {
    const V3 eP = prior.P - pvr.P, eV = prior.V - pvr.V;
    const V3 eR = (prior.R.inverse() * pvr.R).log();
    const V3 eB = (prior.ba + prior.dba) - (bias.ba + bias.dba);
    const V3 all[4] = {eP, eV, eR, eB};
    for (int k = 0; k < 4; k++) { e[3 * k] = all[k].x; e[3 * k + 1] = all[k].y; e[3 * k + 2] = all[k].z; }
}
void edge_prior_jacobians(const NavState& pvr, const double* e, Mat& Jp, Mat& Jb) {
    Jp = Mat(12, 9); Jb = Mat(12, 3);
    Jp.set_block(0, 0, pvr.R.matrix(), -1);
    Jp.set_block(3, 3, M3::identity(), -1);
    Jp.set_block(6, 6, jacobian_r_inv(V3{e[6], e[7], e[8]}));
    Jb.set_block(9, 0, M3::identity(), -1);
}

// ------------------------------------------------------------------------------------------------
// Mini graph + g2o Levenberg-Marquardt
//   optimization_algorithm_levenberg.cpp:61-189, sparse_optimizer.cpp:100-114,354-432,
//   base_{unary,binary,multi}_edge.hpp constructQuadraticForm, robust_kernel_impl.cpp:78-91,
//   base_edge.h:96-102 (rho'' term disabled)
// ------------------------------------------------------------------------------------------------
namespace {
struct Vertex { int dim; bool fixed; bool is_pvr; NavState est; std::vector<NavState> stack; int hidx = -1; };
struct Edge {
    int dim; std::vector<int> v; int level = 0; double delta = 0;        // delta = 0: no robust kernel
    Mat info; std::vector<double> err; std::vector<Mat> J;
    std::function<void(Edge&)> compute_error, linearize;
    double chi2() const { double s = 0; for (int i = 0; i < dim; i++) { double t = 0; for (int j = 0; j < dim; j++) t += info(i, j) * err[j]; s += err[i] * t; } return s; }
};
static void huber(double e, double delta, double* rho) {
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { const double sq = std::sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; rho[2] = -0.5 * rho[1] / e; }
}
struct Graph {
    std::vector<Vertex> V; std::vector<Edge> E;
    int n = 0; Mat H; std::vector<double> b, x;
    double lambda = -1, ni = 2; int nBad = 0;
    std::vector<double> trace; int outer_its = 0; double last_chi = 0;
    bool last_trial_rejected = false;   // the optimize() call in progress ended on a rejected trial: inlier edges keep the rejected state's errors
    void index() { n = 0; for (auto& v : V) { v.hidx = v.fixed ? -1 : n; if (!v.fixed) n += v.dim; } }
    void compute_active_errors() { for (auto& e : E) if (e.level == 0) e.compute_error(e); }
    double active_robust_chi2() {
        double chi = 0;
        for (auto& e : E) if (e.level == 0) { if (e.delta > 0) { double r[3]; huber(e.chi2(), e.delta, r); chi += r[0]; } else chi += e.chi2(); }
        return chi;
    }
    void build_system() {
        H = Mat(n, n); b.assign(n, 0.0);
        for (auto& e : E) {
            if (e.level != 0) continue;
            e.linearize(e);
            double w = 1.0;
            if (e.delta > 0) { double r[3]; huber(e.chi2(), e.delta, r); w = r[1]; }
            std::vector<double> oe(e.dim, 0.0);                         // Omega * e
            for (int i = 0; i < e.dim; i++) for (int j = 0; j < e.dim; j++) oe[i] += e.info(i, j) * e.err[j];
            for (size_t a = 0; a < e.v.size(); a++) {
                const Vertex& va = V[e.v[a]]; if (va.fixed) continue;
                const Mat& Ja = e.J[a];
                for (int c = 0; c < va.dim; c++) { double s = 0; for (int r = 0; r < e.dim; r++) s += Ja(r, c) * oe[r]; b[va.hidx + c] -= w * s; }
                Mat JtO(va.dim, e.dim);                                   // Ja^T * (w Omega)
                for (int c = 0; c < va.dim; c++) for (int r = 0; r < e.dim; r++) { double s = 0; for (int k = 0; k < e.dim; k++) s += Ja(k, c) * e.info(k, r); JtO(c, r) = w * s; }
                for (size_t bb = 0; bb < e.v.size(); bb++) {
                    const Vertex& vb = V[e.v[bb]]; if (vb.fixed) continue;
                    const Mat& Jb = e.J[bb];
                    for (int c = 0; c < va.dim; c++) for (int d = 0; d < vb.dim; d++) { double s = 0; for (int r = 0; r < e.dim; r++) s += JtO(c, r) * Jb(r, d); H(va.hidx + c, vb.hidx + d) += s; }
                }
            }
        }
    }
    void push() { for (auto& v : V) if (!v.fixed) v.stack.push_back(v.est); }
    void pop() { for (auto& v : V) if (!v.fixed) { v.est = v.stack.back(); v.stack.pop_back(); } }
    void discard_top() { for (auto& v : V) if (!v.fixed) v.stack.pop_back(); }
    void update(const std::vector<double>& u) {
        for (auto& v : V) if (!v.fixed) { if (v.is_pvr) v.est.inc_small_pvr(&u[v.hidx]); else v.est.inc_small_bias(&u[v.hidx]); }
    }
    // one OptimizationAlgorithmLevenberg::solve(); returns true for OK, false for Terminate
    bool solve(int iteration) {
        compute_active_errors();
        double currentChi = active_robust_chi2(), tempChi = currentChi;
        const double iniChi = currentChi;
        build_system();
        if (iteration == 0) {
            double maxDiag = 0; for (int i = 0; i < n; i++) maxDiag = std::max(std::fabs(H(i, i)), maxDiag);
            lambda = 1e-5 * maxDiag; ni = 2; nBad = 0;
        }
        double rho = 0; int qmax = 0;
        do {
            push();
            Mat Hl = H; for (int i = 0; i < n; i++) Hl(i, i) += lambda;
            const bool ok2 = cholesky_solve(Hl, b, x);
            if (!ok2) x.assign(n, 0.0);
            update(x);
            compute_active_errors();
            tempChi = active_robust_chi2();
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double scale = 0; for (int j = 0; j < n; j++) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                const double scaleFactor = std::max(1. / 3., alpha);
                lambda *= scaleFactor; ni = 2; currentChi = tempChi; discard_top();
            } else { lambda *= ni; ni *= 2; pop(); }
            qmax++;
        } while (rho < 0 && qmax < 10);
        last_chi = currentChi; trace.push_back(currentChi); outer_its++;
        last_trial_rejected = !(rho > 0 && std::isfinite(tempChi));
        if (qmax == 10 || rho == 0) return false;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) return false;
        return true;
    }
    void optimize(int iterations) {
        index();
        bool ok = true;
        for (int i = 0; i < iterations && ok; i++) ok = solve(i);
    }
};

static double fsqrt(double v) { return (double)(float)std::sqrt(v); }      // "const float th = sqrt(...)"

static Edge make_proj_edge(Graph* g, int vid, const Camera& cam, const Observation& o) {
    Edge e; e.dim = 2; e.v = {vid}; e.delta = fsqrt(5.991); e.info = Mat::identity(2);
    e.info(0, 0) = e.info(1, 1) = o.inv_sigma2; e.err.assign(2, 0.0); e.J.resize(1);
    e.compute_error = [g, vid, cam, o](Edge& s) { edge_proj_error(g->V[vid].est, cam, o, s.err.data()); };
    e.linearize = [g, vid, cam, o](Edge& s) { edge_proj_jacobian(g->V[vid].est, cam, o, s.J[0]); };
    return e;
}
static Edge make_pvr_edge(Graph* g, int vi, int vj, int vb, const Preint& M, V3 gw) {
    Edge e; e.dim = 9; e.v = {vi, vj, vb}; e.delta = fsqrt(21.666); e.err.assign(9, 0.0); e.J.resize(3);
    Mat inv; inverse(M.cov, inv);
    e.info = inv;
    for (int k = 0; k < 3; k++) { e.info(k, k) += 1e2; e.info(3 + k, 3 + k) += 1; e.info(6 + k, 6 + k) += 1e2; }
    e.compute_error = [g, vi, vj, vb, M, gw](Edge& s) { edge_pvr_error(g->V[vi].est, g->V[vj].est, g->V[vb].est, M, gw, s.err.data()); };
    e.linearize = [g, vi, vj, vb, M, gw](Edge& s) { edge_pvr_jacobians(g->V[vi].est, g->V[vj].est, g->V[vb].est, M, gw, s.err.data(), s.J[0], s.J[1], s.J[2]); };
    return e;
}
static Edge make_bias_edge(Graph* g, int vi, int vj, double dt) {
    Edge e; e.dim = 3; e.v = {vi, vj}; e.delta = fsqrt(16.812); e.err.assign(3, 0.0); e.J.resize(2);
    e.info = Mat::identity(3);
    for (int k = 0; k < 3; k++) e.info(k, k) = 1.0 / ImuNoise::accBiasRw2 / dt;
    e.compute_error = [g, vi, vj](Edge& s) {
        const NavState& a = g->V[vi].est; const NavState& b = g->V[vj].est;
        V3 r = (b.ba + b.dba) - (a.ba + a.dba); s.err[0] = r.x; s.err[1] = r.y; s.err[2] = r.z;
    };
    e.linearize = [](Edge& s) { s.J[0] = Mat::identity(3); for (int k = 0; k < 3; k++) s.J[0](k, k) = -1; s.J[1] = Mat::identity(3); };
    return e;
}
static Edge make_prior_edge(Graph* g, int vp, int vb, const NavState& prior, const Mat& margCovInv) {
    Edge e; e.dim = 12; e.v = {vp, vb}; e.delta = fsqrt(30.5779); e.err.assign(12, 0.0); e.J.resize(2);
    e.info = margCovInv;
    for (int k = 0; k < 3; k++) { e.info(k, k) += 1e2; e.info(3 + k, 3 + k) += 1; e.info(6 + k, 6 + k) += 1e2; }
    e.compute_error = [g, vp, vb, prior](Edge& s) { edge_prior_error(g->V[vp].est, g->V[vb].est, prior, s.err.data()); };
    e.linearize = [g, vp](Edge& s) { edge_prior_jacobians(g->V[vp].est, s.err.data(), s.J[0], s.J[1]); };
    return e;
}

// the 4-round outlier scheme shared by both overloads (Optimizer.cc:599-692 / :979-1029)
// Diagnostics of the last run_rounds() on this thread (tests only): rounds whose optimize() ended on a rejected trial — g2o then leaves the
// errors of the REJECTED state on the active edges (optimization_algorithm_levenberg.cpp:143-147 pops the vertices, nothing recomputes
// the errors) and Optimizer.cc:629-634 / :659-664 classify inliers by that stored chi2 —, and edges whose verdict by the stored error
// differs from the verdict by the error at the restored estimate.
thread_local int g_diag_rejected_rounds = 0, g_diag_stale_verdicts = 0;
static int run_rounds(Graph& g, const std::vector<std::pair<int, NavState>>& resets,
                      std::vector<int>& edges_cur, std::vector<uint8_t>& out_cur,
                      std::vector<int>& edges_last, std::vector<uint8_t>& out_last) {
    g_diag_rejected_rounds = 0; g_diag_stale_verdicts = 0;
    const float chi2Mono[4] = {5.991f, 5.991f, 5.991f, 5.991f};
    int nBad = 0;
    for (int it = 0; it < 4; it++) {
        for (auto& r : resets) g.V[r.first].est = r.second;
        g.optimize(10);
        if (g.last_trial_rejected) g_diag_rejected_rounds++;
        auto classify = [&](std::vector<int>& ed, std::vector<uint8_t>& out) {
            int bad = 0;
            for (size_t i = 0; i < ed.size(); i++) {
                Edge& e = g.E[ed[i]];
                if (out[i]) e.compute_error(e);
                else if (g.last_trial_rejected) {                     // diagnostic only: what a recomputation at the restored estimate would say
                    Edge f = e; f.compute_error(f);
                    if (((float)f.chi2() > chi2Mono[it]) != ((float)e.chi2() > chi2Mono[it])) g_diag_stale_verdicts++;
                }
                const float chi2 = (float)e.chi2();
                if (chi2 > chi2Mono[it]) { out[i] = 1; e.level = 1; bad++; } else { out[i] = 0; e.level = 0; }
                if (it == 2) e.delta = 0;
            }
            return bad;
        };
        nBad = classify(edges_cur, out_cur);
        classify(edges_last, out_last);
        if (g.E.size() < 10) break;
    }
    return nBad;
}
} // namespace
void pose_opt_diagnostics(int* rejected_rounds, int* stale_verdicts) { *rejected_rounds = g_diag_rejected_rounds; *stale_verdicts = g_diag_stale_verdicts; }

PoseOptResult pose_opt_vi_kf(const NavState& cur, const NavState& kf, const Preint& preint, V3 gw, const Camera& cam,
                             const std::vector<Observation>& obs, bool marg) {
    PoseOptResult R; R.ns = cur; R.ns_last = kf;
    R.outlier_cur.assign(obs.size(), 0);
    Graph g;
    g.V = {Vertex{9, false, true, cur}, Vertex{3, false, false, cur}, Vertex{9, true, true, kf}, Vertex{3, true, false, kf}};
    g.E.push_back(make_pvr_edge(&g, 2, 0, 3, preint, gw));
    g.E.push_back(make_bias_edge(&g, 3, 1, preint.dt));
    std::vector<int> ec, el; std::vector<uint8_t> ol;
    for (const auto& o : obs) { ec.push_back((int)g.E.size()); g.E.push_back(make_proj_edge(&g, 0, cam, o)); }
    const int nInitial = (int)obs.size();
    if (nInitial < 3) { R.n_inliers = 0; return R; }
    const int nBad = run_rounds(g, {{0, cur}, {1, cur}}, ec, R.outlier_cur, el, ol);
    NavState rec = g.V[0].est; rec.dbg = g.V[1].est.dbg; rec.dba = g.V[1].est.dba;
    R.ns = rec; R.n_inliers = nInitial - nBad; R.final_chi2 = g.last_chi; R.lm_iterations = g.outer_its; R.chi2_trace = g.trace;
    if (marg) {   // block-diagonal inverse of the two diagonal covariance blocks (Optimizer.cc:1069-1090)
        Mat Hinv; inverse(g.H, Hinv);
        Mat c9(9, 9), c3(3, 3), i9, i3;
        for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) c9(i, j) = Hinv(i, j);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c3(i, j) = Hinv(9 + i, 9 + j);
        inverse(c9, i9); inverse(c3, i3);
        R.marg_cov_inv = Mat(12, 12);
        for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) R.marg_cov_inv(i, j) = i9(i, j);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.marg_cov_inv(9 + i, 9 + j) = i3(i, j);
    }
    return R;
}

PoseOptResult pose_opt_vi_frame(const NavState& cur, const NavState& last, const NavState& prior, const Mat& margCovInv,
                                const Preint& preint, V3 gw, const Camera& cam, const std::vector<Observation>& obs_cur,
                                const std::vector<Observation>& obs_last, bool marg) {
    PoseOptResult R; R.ns = cur; R.ns_last = last;
    R.outlier_cur.assign(obs_cur.size(), 0); R.outlier_last.assign(obs_last.size(), 0);
    Graph g;
    g.V = {Vertex{9, false, true, cur}, Vertex{3, false, false, cur}, Vertex{9, false, true, last}, Vertex{3, false, false, last}};
    g.E.push_back(make_prior_edge(&g, 2, 3, prior, margCovInv));
    g.E.push_back(make_pvr_edge(&g, 2, 0, 3, preint, gw));
    g.E.push_back(make_bias_edge(&g, 3, 1, preint.dt));
    std::vector<int> ec, el;
    for (const auto& o : obs_cur) { ec.push_back((int)g.E.size()); g.E.push_back(make_proj_edge(&g, 0, cam, o)); }
    for (const auto& o : obs_last) { el.push_back((int)g.E.size()); g.E.push_back(make_proj_edge(&g, 2, cam, o)); }
    const int nInitial = (int)obs_cur.size();
    if (nInitial < 3) { R.n_inliers = 0; return R; }
    const int nBad = run_rounds(g, {{0, cur}, {1, cur}, {2, last}, {3, last}}, ec, R.outlier_cur, el, R.outlier_last);
    NavState rec = g.V[0].est; rec.dbg = g.V[1].est.dbg; rec.dba = g.V[1].est.dba;
    NavState recl = g.V[2].est; recl.dbg = g.V[3].est.dbg; recl.dba = g.V[3].est.dba;
    R.ns = rec; R.ns_last = recl; R.n_inliers = nInitial - nBad; R.final_chi2 = g.last_chi; R.lm_iterations = g.outer_its; R.chi2_trace = g.trace;
    if (marg) {   // joint 12x12 marginal of (cur PVR, cur bias), inverted (Optimizer.cc:741-768)
        Mat Hinv; inverse(g.H, Hinv);
        Mat c(12, 12);
        for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) c(i, j) = Hinv(i, j);
        inverse(c, R.marg_cov_inv);
    }
    return R;
}

// ------------------------------------------------------------------------------------------------
// Vision-only PoseOptimization(Frame*) — one VertexSE3Expmap, mono (2-D) and stereo (3-D) only-pose edges
// ------------------------------------------------------------------------------------------------
namespace {
struct SE3Q {                                  // g2o::SE3Quat
    Quat r; V3 t;
    void normalize_rotation() { if (r.w < 0) { r.x = -r.x; r.y = -r.y; r.z = -r.z; r.w = -r.w; } r = normalized(r); }
    V3 map(V3 p) const { return rotate(r, p) + t; }
};
static SE3Q se3_mul(const SE3Q& a, const SE3Q& b) { SE3Q o = a; o.t = o.t + rotate(a.r, b.t); o.r = a.r * b.r; o.normalize_rotation(); return o; }
// SE3Quat::exp, se3quat.h:223-257 (update = [omega, upsilon])
static SE3Q se3_exp(const double* u) {
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
    SE3Q o; o.r = quat_from_matrix(R); o.t = Vm * ups; o.normalize_rotation(); return o;
}
struct Se3Edge { Se3Obs o; bool stereo; int level = 0; double delta; double err[3]; double J[3][6]; int dim; };
} // namespace

Se3Result pose_opt_se3(const float* pose12, double fx, double fy, double cx, double cy, double bf, const std::vector<Se3Obs>& obs) {
    Se3Result R; for (int i = 0; i < 12; i++) R.pose12[i] = pose12[i];
    R.outlier.assign(obs.size(), 0);
    auto from_pose = [&]() {                     // Converter::toSE3Quat(pFrame->mTcw)
        M3 Rm; for (int i = 0; i < 9; i++) Rm.m[i] = (double)pose12[i];
        SE3Q s; s.r = quat_from_matrix(Rm); s.t = V3{(double)pose12[9], (double)pose12[10], (double)pose12[11]}; s.normalize_rotation(); return s;
    };
    std::vector<Se3Edge> E(obs.size());
    for (size_t i = 0; i < obs.size(); i++) {
        E[i].o = obs[i]; E[i].stereo = !(obs[i].ur < 0); E[i].dim = E[i].stereo ? 3 : 2;
        E[i].delta = E[i].stereo ? fsqrt(7.815) : fsqrt(5.991);
    }
    const int nInitial = (int)obs.size();
    if (nInitial < 3) { R.n_inliers = 0; return R; }
    SE3Q est = from_pose();
    auto compute_error = [&](Se3Edge& e) {
        const V3 p = est.map(e.o.Xw);
        if (!e.stereo) { e.err[0] = e.o.u - (p.x / p.z * fx + cx); e.err[1] = e.o.v - (p.y / p.z * fy + cy); e.err[2] = 0; }
        else {
            const float invz = (float)(1.0 / p.z);                      // `1.0f/trans_xyz[2]` with a double z: divided in double, rounded once to float (types_six_dof_expmap.cpp:300)
            const double r0 = p.x * invz * fx + cx, r1 = p.y * invz * fy + cy, r2 = r0 - bf * invz;
            e.err[0] = e.o.u - r0; e.err[1] = e.o.v - r1; e.err[2] = e.o.ur - r2;
        }
    };
    auto chi2 = [&](const Se3Edge& e) { double s = 0; for (int k = 0; k < e.dim; k++) s += e.err[k] * e.o.inv_sigma2 * e.err[k]; return s; };
    auto linearize = [&](Se3Edge& e) {
        const V3 p = est.map(e.o.Xw);
        const double x = p.x, y = p.y, invz = 1.0 / p.z, invz_2 = invz * invz;
        double (*J)[6] = e.J;
        J[0][0] = x * y * invz_2 * fx; J[0][1] = -(1 + (x * x * invz_2)) * fx; J[0][2] = y * invz * fx; J[0][3] = -invz * fx; J[0][4] = 0; J[0][5] = x * invz_2 * fx;
        J[1][0] = (1 + y * y * invz_2) * fy; J[1][1] = -x * y * invz_2 * fy; J[1][2] = -x * invz * fy; J[1][3] = 0; J[1][4] = -invz * fy; J[1][5] = y * invz_2 * fy;
        if (e.stereo) { J[2][0] = J[0][0] - bf * y * invz_2; J[2][1] = J[0][1] + bf * x * invz_2; J[2][2] = J[0][2]; J[2][3] = J[0][3]; J[2][4] = 0; J[2][5] = J[0][5] - bf * invz_2; }
    };
    auto active_chi = [&]() { double c = 0; for (auto& e : E) if (e.level == 0) { if (e.delta > 0) { double r[3]; huber(chi2(e), e.delta, r); c += r[0]; } else c += chi2(e); } return c; };
    const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
    int nBad = 0; double lambda = 0, ni = 2; int its_total = 0; double last_chi = 0;
    g_diag_rejected_rounds = 0; g_diag_stale_verdicts = 0;
    for (int round = 0; round < 4; round++) {
        est = from_pose();
        int nBadLM = 0;
        bool last_rejected = false;
        for (int it = 0; it < 10; it++) {
            for (auto& e : E) if (e.level == 0) compute_error(e);
            double currentChi = active_chi(); const double iniChi = currentChi;
            Mat H(6, 6); std::vector<double> b(6, 0.0), x;
            for (auto& e : E) {
                if (e.level != 0) continue;
                linearize(e);
                double w = 1.0; if (e.delta > 0) { double r[3]; huber(chi2(e), e.delta, r); w = r[1]; }
                for (int c = 0; c < 6; c++) {
                    double s = 0; for (int k = 0; k < e.dim; k++) s += e.J[k][c] * e.o.inv_sigma2 * e.err[k];
                    b[c] -= w * s;
                    for (int d = 0; d < 6; d++) { double q = 0; for (int k = 0; k < e.dim; k++) q += e.J[k][c] * e.o.inv_sigma2 * e.J[k][d]; H(c, d) += w * q; }
                }
            }
            if (it == 0) { double mx = 0; for (int i = 0; i < 6; i++) mx = std::max(std::fabs(H(i, i)), mx); lambda = 1e-5 * mx; ni = 2; nBadLM = 0; }
            double rho = 0; int qmax = 0;
            do {
                const SE3Q backup = est;
                Mat Hl = H; for (int i = 0; i < 6; i++) Hl(i, i) += lambda;
                const bool ok2 = cholesky_solve(Hl, b, x);
                if (!ok2) x.assign(6, 0.0);
                est = se3_mul(se3_exp(x.data()), est);                 // VertexSE3Expmap::oplusImpl
                for (auto& e : E) if (e.level == 0) compute_error(e);
                double tempChi = active_chi();
                if (!ok2) tempChi = std::numeric_limits<double>::max();
                double scale = 0; for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                scale += 1e-3;
                rho = (currentChi - tempChi) / scale;
                if (rho > 0 && std::isfinite(tempChi)) {
                    double alpha = 1. - std::pow((2 * rho - 1), 3); alpha = std::min(alpha, 2. / 3.);
                    lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi;
                    last_rejected = false;
                } else { lambda *= ni; ni *= 2; est = backup; last_rejected = true; }
                qmax++;
            } while (rho < 0 && qmax < 10);
            its_total++; last_chi = currentChi;
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
            if (nBadLM >= 3) break;
        }
        nBad = 0;
        if (last_rejected) g_diag_rejected_rounds++;
        for (size_t i = 0; i < E.size(); i++) {
            Se3Edge& e = E[i];
            if (R.outlier[i]) compute_error(e);            // an inlier keeps the error of the last computeActiveErrors (the last trial state, also when it was rejected)
            else if (last_rejected) { Se3Edge f = e; compute_error(f); if (((float)chi2(f) > (e.stereo ? chi2Stereo : chi2Mono)) != ((float)chi2(e) > (e.stereo ? chi2Stereo : chi2Mono))) g_diag_stale_verdicts++; }
            const float c2 = (float)chi2(e);
            if (c2 > (e.stereo ? chi2Stereo : chi2Mono)) { R.outlier[i] = 1; e.level = 1; nBad++; } else { R.outlier[i] = 0; e.level = 0; }
            if (round == 2) e.delta = 0;
        }
        if (E.size() < 10) break;
    }
    const M3 Rm = to_matrix(est.r);
    for (int i = 0; i < 9; i++) R.pose12[i] = (float)Rm.m[i];
    R.pose12[9] = (float)est.t.x; R.pose12[10] = (float)est.t.y; R.pose12[11] = (float)est.t.z;
    R.n_inliers = nInitial - nBad; R.final_chi2 = last_chi; R.lm_iterations = its_total;
    return R;
}

} // namespace ora

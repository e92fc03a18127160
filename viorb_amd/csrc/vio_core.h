// viorb_amd/csrc/vio_core.h — FP64 building blocks of the visual-inertial solve, shared by the HIP
// kernels (vio_kernels.hip) and by host-only debug hooks that let the CPU test-suite compare them
// with the oracle without a GPU. Fixed-size values live in registers (static indexing only).
//
// What is restated (reference file:line):
//   SO3 exp/log/JacobianR/JacobianRInv          src/IMU/so3.cpp:32-100,198-280
//   NavState::IncSmallPVR / IncSmallBias        src/IMU/NavState.cpp:71-140
//   IMUPreintegrator::update                    src/IMU/IMUPreintegrator.cpp:86-153
//   Converter::updateNS                         src/Converter.cc:27-49
//   EdgeNavStatePVR error + Jacobians           src/IMU/g2otypes.cpp:8-229
//   EdgeNavStatePVRPointXYZOnlyPose             src/IMU/g2otypes.h:205-281, g2otypes.cpp:356-407
//   EdgeNavStatePriorPVRBias (12-D)             src/IMU/g2otypes.cpp:409-515
//   EdgeNavStateBias (3-D)                      src/IMU/g2otypes.cpp:231-297
//   Frame::UpdatePoseFromNS                     src/Frame.cc:88-105
// Flat layouts (include/viorb.h): navstate[22] = P3 V3 q4(x,y,z,w) bg3 ba3 dbg3 dba3;
// preint[142] = dP3 dV3 dR9 JPg9 JPa9 JVg9 JVa9 JRg9 cov81 dt; cam[16] = fx fy cx cy Rbc9 Pbc3.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VIO_HD __host__ __device__ __forceinline__
#else
#define VIO_HD inline
#endif

namespace viorb {

struct d3 { double x, y, z; };
VIO_HD d3 mk3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
VIO_HD d3 ld3(const double* p) { return mk3(p[0], p[1], p[2]); }
VIO_HD void st3(double* p, d3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
VIO_HD d3 operator+(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VIO_HD d3 operator-(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VIO_HD d3 operator*(d3 a, double s) { return mk3(a.x * s, a.y * s, a.z * s); }
VIO_HD double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VIO_HD d3 cross3(d3 a, d3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
VIO_HD double norm3(d3 a) { return sqrt(dot3(a, a)); }

struct m33 { double a00, a01, a02, a10, a11, a12, a20, a21, a22; };
VIO_HD m33 mkm(double a00, double a01, double a02, double a10, double a11, double a12, double a20, double a21, double a22) {
    m33 m; m.a00 = a00; m.a01 = a01; m.a02 = a02; m.a10 = a10; m.a11 = a11; m.a12 = a12; m.a20 = a20; m.a21 = a21; m.a22 = a22; return m;
}
VIO_HD m33 eye3() { return mkm(1, 0, 0, 0, 1, 0, 0, 0, 1); }
VIO_HD m33 zero3() { return mkm(0, 0, 0, 0, 0, 0, 0, 0, 0); }
VIO_HD m33 ldm(const double* p) { return mkm(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]); }
VIO_HD void stm(double* p, const m33& m) { p[0] = m.a00; p[1] = m.a01; p[2] = m.a02; p[3] = m.a10; p[4] = m.a11; p[5] = m.a12; p[6] = m.a20; p[7] = m.a21; p[8] = m.a22; }
VIO_HD m33 mul(const m33& a, const m33& b) {
    return mkm(a.a00 * b.a00 + a.a01 * b.a10 + a.a02 * b.a20, a.a00 * b.a01 + a.a01 * b.a11 + a.a02 * b.a21, a.a00 * b.a02 + a.a01 * b.a12 + a.a02 * b.a22,
               a.a10 * b.a00 + a.a11 * b.a10 + a.a12 * b.a20, a.a10 * b.a01 + a.a11 * b.a11 + a.a12 * b.a21, a.a10 * b.a02 + a.a11 * b.a12 + a.a12 * b.a22,
               a.a20 * b.a00 + a.a21 * b.a10 + a.a22 * b.a20, a.a20 * b.a01 + a.a21 * b.a11 + a.a22 * b.a21, a.a20 * b.a02 + a.a21 * b.a12 + a.a22 * b.a22);
}
VIO_HD d3 mulv(const m33& a, d3 v) { return mk3(a.a00 * v.x + a.a01 * v.y + a.a02 * v.z, a.a10 * v.x + a.a11 * v.y + a.a12 * v.z, a.a20 * v.x + a.a21 * v.y + a.a22 * v.z); }
VIO_HD m33 tr(const m33& a) { return mkm(a.a00, a.a10, a.a20, a.a01, a.a11, a.a21, a.a02, a.a12, a.a22); }
VIO_HD m33 scl(const m33& a, double s) { return mkm(a.a00 * s, a.a01 * s, a.a02 * s, a.a10 * s, a.a11 * s, a.a12 * s, a.a20 * s, a.a21 * s, a.a22 * s); }
VIO_HD m33 add(const m33& a, const m33& b) { return mkm(a.a00 + b.a00, a.a01 + b.a01, a.a02 + b.a02, a.a10 + b.a10, a.a11 + b.a11, a.a12 + b.a12, a.a20 + b.a20, a.a21 + b.a21, a.a22 + b.a22); }
VIO_HD m33 sub(const m33& a, const m33& b) { return add(a, scl(b, -1.0)); }
VIO_HD m33 hat3(d3 v) { return mkm(0, -v.z, v.y, v.z, 0, -v.x, -v.y, v.x, 0); }

struct quat { double x, y, z, w; };
VIO_HD quat mkq(double x, double y, double z, double w) { quat q; q.x = x; q.y = y; q.z = z; q.w = w; return q; }
VIO_HD quat qnorm(quat q) { const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w); return mkq(q.x / n, q.y / n, q.z / n, q.w / n); }
VIO_HD quat qmul(quat a, quat b) {
    return mkq(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
               a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
VIO_HD quat qconj(quat q) { return mkq(-q.x, -q.y, -q.z, q.w); }
// rotation matrix <-> quaternion (Eigen 3 conventions; coefficient order x, y, z, w)
VIO_HD m33 qmat(quat q) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    return mkm(1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy));
}
VIO_HD quat mat2q(const m33& m) {
    const double t = m.a00 + m.a11 + m.a22;
    if (t > 0) { const double s = sqrt(t + 1.0), r = 0.5 / s; return mkq((m.a21 - m.a12) * r, (m.a02 - m.a20) * r, (m.a10 - m.a01) * r, 0.5 * s); }
    if (m.a00 >= m.a11 && m.a00 >= m.a22) { const double s = sqrt(m.a00 - m.a11 - m.a22 + 1.0), r = 0.5 / s; return mkq(0.5 * s, (m.a10 + m.a01) * r, (m.a20 + m.a02) * r, (m.a21 - m.a12) * r); }
    if (m.a11 > m.a00 && m.a11 >= m.a22) { const double s = sqrt(m.a11 - m.a22 - m.a00 + 1.0), r = 0.5 / s; return mkq((m.a10 + m.a01) * r, 0.5 * s, (m.a21 + m.a12) * r, (m.a02 - m.a20) * r); }
    const double s = sqrt(m.a22 - m.a00 - m.a11 + 1.0), r = 0.5 / s; return mkq((m.a02 + m.a20) * r, (m.a21 + m.a12) * r, 0.5 * s, (m.a10 - m.a01) * r);
}
VIO_HD d3 qrot(quat q, d3 v) { d3 qv = mk3(q.x, q.y, q.z); d3 uv = cross3(qv, v); uv = uv + uv; return v + uv * q.w + cross3(qv, uv); }
VIO_HD quat so3_mul(quat a, quat b) { return qnorm(qmul(qnorm(a), b)); }      // Sophus::SO3::operator*
// so3.cpp expAndTheta (SMALL_EPS 1e-10)
VIO_HD quat so3_exp(d3 w) {
    const double th = norm3(w), half = 0.5 * th;
    double imag; const double real = cos(half);
    if (th < 1e-10) { const double t2 = th * th, t4 = t2 * t2; imag = 0.5 - 0.0208333 * t2 + 0.000260417 * t4; }
    else imag = sin(half) / th;
    return qnorm(mkq(imag * w.x, imag * w.y, imag * w.z, real));
}
// so3.cpp logAndTheta: the |w| < eps branch is overwritten by the atan form in the reference
VIO_HD d3 so3_log(quat q) {
    const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z), w = q.w;
    const double f = (n < 1e-10) ? (2. / w - 2. * (n * n) / (w * w * w)) : (2 * atan(n / w) / n);
    return mk3(f * q.x, f * q.y, f * q.z);
}
VIO_HD m33 so3_jr(d3 w) {
    const double th = norm3(w);
    if (th < 0.00001) return eye3();
    const m33 K = hat3(w * (1.0 / th));
    return add(sub(eye3(), scl(K, (1 - cos(th)) / th)), scl(mul(K, K), 1 - sin(th) / th));
}
VIO_HD m33 so3_jr_inv(d3 w) {
    const double th = norm3(w);
    if (th < 0.00001) return eye3();
    const m33 K = hat3(w * (1.0 / th));
    return add(add(eye3(), scl(hat3(w), 0.5)), scl(mul(K, K), 1.0 - (1.0 + cos(th)) * th / (2.0 * sin(th))));
}

// PVR part of a NavState as the solver carries it
struct pvr { d3 P, V; quat q; };
VIO_HD pvr ld_pvr(const double* ns) { pvr s; s.P = ld3(ns); s.V = ld3(ns + 3); s.q = qnorm(mkq(ns[6], ns[7], ns[8], ns[9])); return s; }
VIO_HD void st_pvr(double* ns, const pvr& s) { st3(ns, s.P); st3(ns + 3, s.V); ns[6] = s.q.x; ns[7] = s.q.y; ns[8] = s.q.z; ns[9] = s.q.w; }
// NavState::IncSmallPVR
VIO_HD pvr inc_small_pvr(const pvr& s, const double* u) {
    pvr r;
    r.P = s.P + mulv(qmat(s.q), mk3(u[0], u[1], u[2]));
    r.V = s.V + mk3(u[3], u[4], u[5]);
    r.q = so3_mul(s.q, so3_exp(mk3(u[6], u[7], u[8])));
    return r;
}

// ---- IMU pre-integration: everything of update() except the 9x9 covariance product, which the
// kernel distributes over lanes (A, Bg, Ca blocks are returned for it).
struct preint_small { d3 dP, dV; m33 dR, JPg, JPa, JVg, JVa, JRg; double dt; };
struct preint_cov_blocks { m33 A66, A36, A06, Bg, Ca3, Ca0; double dt; };   // A(6,6) A(3,6) A(0,6); Bg(6,0); Ca(3,0) Ca(0,0); A(0,3)=I*dt
VIO_HD m33 normalize_rotation(const m33& R) {
    quat q = mat2q(R);
    if (q.w < 0) q = mkq(-q.x, -q.y, -q.z, -q.w);
    return qmat(qnorm(q));
}
VIO_HD preint_cov_blocks preint_step(preint_small& M, d3 omega, d3 acc, double d) {
    const double dt2 = d * d;
    const m33 dRk = qmat(so3_exp(omega * d));
    const m33 Jr = so3_jr(omega * d);
    const m33 RA = mul(M.dR, hat3(acc));
    preint_cov_blocks C;
    C.A66 = tr(dRk); C.A36 = scl(RA, -d); C.A06 = scl(RA, -0.5 * dt2);
    C.Bg = scl(Jr, d); C.Ca3 = scl(M.dR, d); C.Ca0 = scl(M.dR, 0.5 * dt2); C.dt = d;
    const m33 RAJ = mul(RA, M.JRg);
    M.JPa = sub(add(M.JPa, scl(M.JVa, d)), scl(M.dR, 0.5 * dt2));
    M.JPg = sub(add(M.JPg, scl(M.JVg, d)), scl(RAJ, 0.5 * dt2));
    M.JVa = sub(M.JVa, scl(M.dR, d));
    M.JVg = sub(M.JVg, scl(RAJ, d));
    M.JRg = sub(mul(tr(dRk), M.JRg), scl(Jr, d));
    const d3 Ra = mulv(M.dR, acc);
    M.dP = M.dP + M.dV * d + Ra * (0.5 * dt2);
    M.dV = M.dV + Ra * d;
    M.dR = normalize_rotation(mul(M.dR, dRk));
    M.dt += d;
    return C;
}
// element (r, c) of the 9x9 matrices A, and of the noise terms Bg*Sg*Bg^T + Ca*Sa*Ca^T
VIO_HD double m33_at(const m33& m, int r, int c) {
    const double v[9] = {m.a00, m.a01, m.a02, m.a10, m.a11, m.a12, m.a20, m.a21, m.a22};
    return v[3 * r + c];
}

// Converter::updateNS
VIO_HD pvr update_ns(const pvr& s, d3 dP, d3 dV, const m33& dR, double dt, d3 gw) {
    const m33 Rw = qmat(s.q);
    pvr r;
    r.P = s.P + s.V * dt + ((gw * 0.5) * dt) * dt + mulv(Rw, dP);
    r.V = s.V + gw * dt + mulv(Rw, dV);
    r.q = qnorm(mat2q(mul(Rw, dR)));
    return r;
}

// Frame::UpdatePoseFromNS in the float arithmetic of its cv::Mat expressions: Rwb, Pwb, Rbc, Pbc are
// rounded to float first; Rcw = (Rwb*Rbc)^T; Pwc = Rwb*Pbc + Pwb; Pcw = -Rcw*Pwc (3-term float sums,
// left to right, as cv::gemm's small-matrix path does). pose12 = Rcw (row-major) + tcw.
VIO_HD void pose_from_navstate_f32(const pvr& s, const double* cam16, float* pose12) {
    const m33 Rd = qmat(s.q);
    const float Rwb[9] = {(float)Rd.a00, (float)Rd.a01, (float)Rd.a02, (float)Rd.a10, (float)Rd.a11, (float)Rd.a12, (float)Rd.a20, (float)Rd.a21, (float)Rd.a22};
    float Rbc[9], Pbc[3];
    for (int i = 0; i < 9; i++) Rbc[i] = (float)cam16[4 + i];
    for (int i = 0; i < 3; i++) Pbc[i] = (float)cam16[13 + i];
    const float Pwb[3] = {(float)s.P.x, (float)s.P.y, (float)s.P.z};
    float Rwc[9], Pwc[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Rwc[3 * r + c] = Rwb[3 * r] * Rbc[c] + Rwb[3 * r + 1] * Rbc[3 + c] + Rwb[3 * r + 2] * Rbc[6 + c];
        const float t = Rwb[3 * r] * Pbc[0] + Rwb[3 * r + 1] * Pbc[1] + Rwb[3 * r + 2] * Pbc[2];
        Pwc[r] = t + Pwb[r];
    }
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) pose12[3 * r + c] = Rwc[3 * c + r];
        // (-Rcw) * Pwc : the negation is applied to Rcw's entries before the products
        const float t = (-Rwc[r]) * Pwc[0] + (-Rwc[3 + r]) * Pwc[1] + (-Rwc[6 + r]) * Pwc[2];
        pose12[9 + r] = t;
    }
}

// ---- reprojection edge (EdgeNavStatePVRPointXYZOnlyPose) -------------------------------------
struct cam_t { double fx, fy, cx, cy; m33 Rcb; d3 RcbPbc; };
VIO_HD cam_t ld_cam(const double* c) { cam_t k; k.fx = c[0]; k.fy = c[1]; k.cx = c[2]; k.cy = c[3]; k.Rcb = tr(ldm(c + 4)); k.RcbPbc = mulv(k.Rcb, ld3(c + 13)); return k; }
// e = obs - proj(Pc); J = d e / d (dP, dPhi) as 2x3 | 2x3 (the velocity block is zero)
VIO_HD void proj_edge(const cam_t& k, const m33& RwbT, d3 Pwb, d3 Pw, double u, double v, bool jac, double* e, double* JP, double* JR) {
    const d3 Paux = mulv(k.Rcb, mulv(RwbT, Pw - Pwb));
    const d3 Pc = Paux - k.RcbPbc;
    const double iz = 1.0 / Pc.z, xz = Pc.x * iz, yz = Pc.y * iz;
    e[0] = u - (xz * k.fx + k.cx);
    e[1] = v - (yz * k.fy + k.cy);
    if (!jac) return;
    const double j00 = k.fx * iz, j02 = -xz * k.fx * iz, j11 = k.fy * iz, j12 = -yz * k.fy * iz;
    const m33 HR = mul(hat3(Paux), k.Rcb);
    // JdPwb = Jpi * Rcb ; JdRwb = -Jpi * (hat(Paux) * Rcb)
    JP[0] = j00 * k.Rcb.a00 + j02 * k.Rcb.a20; JP[1] = j00 * k.Rcb.a01 + j02 * k.Rcb.a21; JP[2] = j00 * k.Rcb.a02 + j02 * k.Rcb.a22;
    JP[3] = j11 * k.Rcb.a10 + j12 * k.Rcb.a20; JP[4] = j11 * k.Rcb.a11 + j12 * k.Rcb.a21; JP[5] = j11 * k.Rcb.a12 + j12 * k.Rcb.a22;
    JR[0] = -(j00 * HR.a00 + j02 * HR.a20); JR[1] = -(j00 * HR.a01 + j02 * HR.a21); JR[2] = -(j00 * HR.a02 + j02 * HR.a22);
    JR[3] = -(j11 * HR.a10 + j12 * HR.a20); JR[4] = -(j11 * HR.a11 + j12 * HR.a21); JR[5] = -(j11 * HR.a12 + j12 * HR.a22);
}

// ---- g2o::SE3Quat for the vision-only pose solve (Thirdparty/g2o/g2o/types/se3quat.h) ---------------------
struct se3q { quat r; d3 t; };
VIO_HD quat se3_norm_rot(quat r) { if (r.w < 0) r = mkq(-r.x, -r.y, -r.z, -r.w); return qnorm(r); }
VIO_HD d3 se3_map(const se3q& s, d3 p) { return qrot(s.r, p) + s.t; }
VIO_HD se3q se3_mul(const se3q& a, const se3q& b) { se3q o; o.t = a.t + qrot(a.r, b.t); o.r = se3_norm_rot(qmul(a.r, b.r)); return o; }
// SE3Quat::exp(update = [omega, upsilon]), se3quat.h:223-257
VIO_HD se3q se3_exp(const double* u) {
    const d3 om = mk3(u[0], u[1], u[2]), ups = mk3(u[3], u[4], u[5]);
    const double th = norm3(om);
    const m33 Om = hat3(om), Om2 = mul(Om, Om);
    m33 R, V;
    if (th < 0.00001) { R = add(add(eye3(), Om), Om2); V = R; }
    else {
        const double a = sin(th) / th, b = (1 - cos(th)) / (th * th), c = (th - sin(th)) / (th * th * th);
        R = add(add(eye3(), scl(Om, a)), scl(Om2, b));
        V = add(add(eye3(), scl(Om, b)), scl(Om2, c));
    }
    se3q o; o.r = se3_norm_rot(mat2q(R)); o.t = mulv(V, ups); return o;
}
// EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose (types_six_dof_expmap.cpp:266-364): error (dim 2 or 3)
// and the 6-column Jacobian rows; the stereo projection uses a FLOAT reciprocal depth, as the reference does.
VIO_HD int se3_edge(const se3q& s, d3 Xw, double u, double v, double ur, double fx, double fy, double cx, double cy, double bf,
                    bool jac, double* e, double* J /* 3 x 6 */) {
    const d3 p = se3_map(s, Xw);
    const bool stereo = !(ur < 0);
    if (!stereo) { e[0] = u - (p.x / p.z * fx + cx); e[1] = v - (p.y / p.z * fy + cy); e[2] = 0; }
    else {
        const float invzf = (float)(1.0 / p.z);                     // `1.0f/trans_xyz[2]`: double division rounded once to float (types_six_dof_expmap.cpp:300)
        const double r0 = p.x * invzf * fx + cx, r1 = p.y * invzf * fy + cy, r2 = r0 - bf * invzf;
        e[0] = u - r0; e[1] = v - r1; e[2] = ur - r2;
    }
    if (jac) {
        const double x = p.x, y = p.y, invz = 1.0 / p.z, invz_2 = invz * invz;
        J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx; J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
        J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy; J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
        if (stereo) { J[12] = J[0] - bf * y * invz_2; J[13] = J[1] + bf * x * invz_2; J[14] = J[2]; J[15] = J[3]; J[16] = 0; J[17] = J[5] - bf * invz_2; }
        else { for (int k = 12; k < 18; k++) J[k] = 0; }
    }
    return stereo ? 3 : 2;
}

// RobustKernelHuber::robustify (rho, rho')
VIO_HD void huber(double e, double delta, double* rho0, double* rho1) {
    const double dsqr = delta * delta;
    if (e <= dsqr) { *rho0 = e; *rho1 = 1.0; }
    else { const double sq = sqrt(e); *rho0 = 2 * sq * delta - dsqr; *rho1 = delta / sq; }
}

// ---- IMU factor (EdgeNavStatePVR): error e[9]; Jacobians written as dense row-major 9 x 21 =
// [ d/d(i: P V Phi) | d/d(j: P V Phi) | d/d(bias_i acc) ]
VIO_HD void pvr_edge(const pvr& si, const pvr& sj, d3 dbg_i, d3 dba_i, const double* preint, d3 gw, double* e, double* J /* 9*21 or null */,
                     bool zero_fill = true) {
    const d3 dP = ld3(preint), dV = ld3(preint + 3);
    const m33 dR = ldm(preint + 6), JPg = ldm(preint + 15), JPa = ldm(preint + 24), JVg = ldm(preint + 33), JVa = ldm(preint + 42), JRg = ldm(preint + 51);
    const double dT = preint[141], dT2 = dT * dT;
    const quat RiT = qnorm(qconj(si.q));
    const d3 aP = qrot(RiT, sj.P - si.P - si.V * dT - gw * (0.5 * dT2));
    const d3 aV = qrot(RiT, sj.V - si.V - gw * dT);
    const d3 rP = aP - (dP + mulv(JPg, dbg_i) + mulv(JPa, dba_i));
    const d3 rV = aV - (dV + mulv(JVg, dbg_i) + mulv(JVa, dba_i));
    const quat dRij = qnorm(mat2q(dR));
    const quat corr = so3_mul(dRij, so3_exp(mulv(JRg, dbg_i)));
    const quat rR = so3_mul(so3_mul(qnorm(qconj(corr)), RiT), sj.q);
    const d3 rPhi = so3_log(rR);
    e[0] = rP.x; e[1] = rP.y; e[2] = rP.z; e[3] = rV.x; e[4] = rV.y; e[5] = rV.z; e[6] = rPhi.x; e[7] = rPhi.y; e[8] = rPhi.z;
    if (!J) return;
    if (zero_fill) for (int i = 0; i < 9 * 21; i++) J[i] = 0;
    const m33 Ri = qmat(si.q), Rj = qmat(sj.q), RiTm = tr(Ri), RjT = tr(Rj);
    const m33 JrInv = so3_jr_inv(rPhi);
    auto put = [&](int r0, int c0, const m33& B, double s) {
        const double v[9] = {B.a00, B.a01, B.a02, B.a10, B.a11, B.a12, B.a20, B.a21, B.a22};
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) J[(r0 + r) * 21 + c0 + c] = s * v[3 * r + c];
    };
    put(0, 0, eye3(), -1); put(0, 3, RiTm, -dT); put(0, 6, hat3(aP), 1);
    put(3, 3, RiTm, -1); put(3, 6, hat3(aV), 1);
    put(6, 6, mul(mul(JrInv, RjT), Ri), -1);
    put(0, 9, mul(RiTm, Rj), 1); put(3, 12, RiTm, 1); put(6, 15, JrInv, 1);
    put(0, 18, JPa, -1); put(3, 18, JVa, -1);
}
// ---- prior factor (EdgeNavStatePriorPVRBias, 12-D): J row-major 12 x 12 = [ d/d(P V Phi) | d/d(bias acc) ]
VIO_HD void prior_edge(const pvr& s, d3 ba_plus_dba, const double* prior22, double* e, double* J /* 12*12 or null */, bool zero_fill = true) {
    const pvr pr = ld_pvr(prior22);
    const d3 eP = pr.P - s.P, eV = pr.V - s.V;
    const d3 eR = so3_log(so3_mul(qnorm(qconj(pr.q)), s.q));
    const d3 eB = (ld3(prior22 + 13) + ld3(prior22 + 19)) - ba_plus_dba;
    e[0] = eP.x; e[1] = eP.y; e[2] = eP.z; e[3] = eV.x; e[4] = eV.y; e[5] = eV.z; e[6] = eR.x; e[7] = eR.y; e[8] = eR.z; e[9] = eB.x; e[10] = eB.y; e[11] = eB.z;
    if (!J) return;
    if (zero_fill) for (int i = 0; i < 144; i++) J[i] = 0;
    const m33 R = qmat(s.q), Ji = so3_jr_inv(eR);
    const double rv[9] = {R.a00, R.a01, R.a02, R.a10, R.a11, R.a12, R.a20, R.a21, R.a22};
    const double jv[9] = {Ji.a00, Ji.a01, Ji.a02, Ji.a10, Ji.a11, Ji.a12, Ji.a20, Ji.a21, Ji.a22};
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { J[r * 12 + c] = -rv[3 * r + c]; J[(6 + r) * 12 + 6 + c] = jv[3 * r + c]; }
    for (int r = 0; r < 3; r++) { J[(3 + r) * 12 + 3 + r] = -1; J[(9 + r) * 12 + 9 + r] = -1; }
}

} // namespace viorb

// oracle/vio_math.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Small dense FP64 algebra + SO(3) as the reference uses them through Eigen3 / its Sophus fork
// (reference src/IMU/so3.{h,cpp}; Eigen is NOT vendored: Quaternion<->matrix conversions follow
// Eigen 3's published algorithms). FP64 results are compared with 1e-5-relative tolerances, so
// operation order inside these helpers is not part of the parity contract.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace ora {

struct V3 { double x = 0, y = 0, z = 0; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return a * s; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

struct M3 {                      // row-major
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double& operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }
    static M3 identity() { M3 a; a.m[0] = a.m[4] = a.m[8] = 1; return a; }
};
inline M3 operator*(const M3& a, const M3& b) {
    M3 c;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += a(i, k) * b(k, j); c(i, j) = s; }
    return c;
}
inline V3 operator*(const M3& a, V3 v) {
    return {a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z, a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z,
            a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z};
}
inline M3 operator*(const M3& a, double s) { M3 c; for (int i = 0; i < 9; i++) c.m[i] = a.m[i] * s; return c; }
inline M3 operator+(const M3& a, const M3& b) { M3 c; for (int i = 0; i < 9; i++) c.m[i] = a.m[i] + b.m[i]; return c; }
inline M3 operator-(const M3& a, const M3& b) { M3 c; for (int i = 0; i < 9; i++) c.m[i] = a.m[i] - b.m[i]; return c; }
inline M3 operator-(const M3& a) { return a * -1.0; }
inline M3 transpose(const M3& a) { M3 c; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c(i, j) = a(j, i); return c; }
// SO3::hat, reference src/IMU/so3.cpp (hat)
inline M3 hat(V3 v) { M3 o; o(0, 1) = -v.z; o(0, 2) = v.y; o(1, 0) = v.z; o(1, 2) = -v.x; o(2, 0) = -v.y; o(2, 1) = v.x; return o; }

// Unit quaternion with Eigen's coefficient order (x, y, z, w) and Eigen 3's conversion algorithms.
struct Quat { double x = 0, y = 0, z = 0, w = 1; };
inline Quat normalized(Quat q) { double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w); return {q.x / n, q.y / n, q.z / n, q.w / n}; }
inline Quat operator*(Quat a, Quat b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Quat conj(Quat q) { return {-q.x, -q.y, -q.z, q.w}; }
inline Quat quat_from_matrix(const M3& m) {          // Eigen::Quaterniond(Matrix3d)
    Quat q; double c[4];
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0) {
        t = std::sqrt(t + 1.0); q.w = 0.5 * t; t = 0.5 / t;
        q.x = (m(2, 1) - m(1, 2)) * t; q.y = (m(0, 2) - m(2, 0)) * t; q.z = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0; if (m(1, 1) > m(0, 0)) i = 1; if (m(2, 2) > m(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        c[i] = 0.5 * t; t = 0.5 / t;
        c[3] = (m(k, j) - m(j, k)) * t; c[j] = (m(j, i) + m(i, j)) * t; c[k] = (m(k, i) + m(i, k)) * t;
        q.x = c[0]; q.y = c[1]; q.z = c[2]; q.w = c[3];
    }
    return q;
}
inline M3 to_matrix(Quat q) {                        // Eigen toRotationMatrix
    M3 r;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    r(0, 0) = 1 - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
    r(1, 0) = txy + twz; r(1, 1) = 1 - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1 - (txx + tyy);
    return r;
}
inline V3 rotate(Quat q, V3 v) {                     // Eigen _transformVector
    V3 qv{q.x, q.y, q.z};
    V3 uv = cross(qv, v); uv = uv + uv;
    return v + uv * q.w + cross(qv, uv);
}

// Sophus::SO3 of the reference (unit quaternion inside; every product re-normalises)
struct SO3 {
    Quat q;
    SO3() {}
    explicit SO3(Quat qq) : q(normalized(qq)) {}
    explicit SO3(const M3& R) : q(normalized(quat_from_matrix(R))) {}
    M3 matrix() const { return to_matrix(q); }
    SO3 inverse() const { return SO3(conj(q)); }
    SO3 operator*(const SO3& o) const { return SO3(normalized(q) * o.q); }
    V3 operator*(V3 v) const { return rotate(q, v); }
    // reference src/IMU/so3.cpp expAndTheta: SMALL_EPS = 1e-10
    static SO3 exp(V3 omega) {
        const double theta = norm(omega), half = 0.5 * theta;
        double imag; const double real = std::cos(half);
        if (theta < 1e-10) { double t2 = theta * theta, t4 = t2 * t2; imag = 0.5 - 0.0208333 * t2 + 0.000260417 * t4; }
        else imag = std::sin(half) / theta;
        return SO3(Quat{imag * omega.x, imag * omega.y, imag * omega.z, real});
    }
    // reference logAndTheta: note the |w| < eps branch is overwritten by the atan form (no else)
    V3 log() const {
        const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z), w = q.w;
        double f;
        if (n < 1e-10) f = 2. / w - 2. * (n * n) / (w * w * w);
        else f = 2 * std::atan(n / w) / n;
        return {f * q.x, f * q.y, f * q.z};
    }
};
// reference src/IMU/so3.cpp JacobianR / JacobianRInv (threshold 1e-5)
inline M3 jacobian_r(V3 w) {
    M3 J = M3::identity();
    const double th = norm(w);
    if (th < 0.00001) return J;
    V3 k = w * (1.0 / th); M3 K = hat(k);
    return M3::identity() - K * ((1 - std::cos(th)) / th) + (K * K) * (1 - std::sin(th) / th);
}
inline M3 jacobian_r_inv(V3 w) {
    M3 J = M3::identity();
    const double th = norm(w);
    if (th < 0.00001) return J;
    V3 k = w * (1.0 / th); M3 K = hat(k);
    return M3::identity() + hat(w) * 0.5 + (K * K) * (1.0 - (1.0 + std::cos(th)) * th / (2.0 * std::sin(th)));
}

// ---- runtime-sized dense matrices (row-major) ------------------------------------------------
struct Mat {
    int r = 0, c = 0; std::vector<double> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
    static Mat identity(int n) { Mat m(n, n); for (int i = 0; i < n; i++) m(i, i) = 1; return m; }
    void set_block(int i0, int j0, const M3& b, double s = 1.0) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) (*this)(i0 + i, j0 + j) = s * b(i, j); }
};
inline Mat operator*(const Mat& x, const Mat& y) {
    Mat z(x.r, y.c);
    for (int i = 0; i < x.r; i++) for (int k = 0; k < x.c; k++) { double v = x(i, k); if (v == 0) continue; for (int j = 0; j < y.c; j++) z(i, j) += v * y(k, j); }
    return z;
}
inline Mat operator+(const Mat& x, const Mat& y) { Mat z = x; for (size_t i = 0; i < z.a.size(); i++) z.a[i] += y.a[i]; return z; }
inline Mat transpose(const Mat& x) { Mat z(x.c, x.r); for (int i = 0; i < x.r; i++) for (int j = 0; j < x.c; j++) z(j, i) = x(i, j); return z; }
// General inverse by LU with partial pivoting (what Eigen's .inverse() does for n > 4).
inline bool inverse(const Mat& x, Mat& out) {
    const int n = x.r; Mat a = x; out = Mat::identity(n);
    for (int col = 0; col < n; col++) {
        int p = col; double best = std::fabs(a(col, col));
        for (int i = col + 1; i < n; i++) if (std::fabs(a(i, col)) > best) { best = std::fabs(a(i, col)); p = i; }
        if (best == 0) return false;
        if (p != col) for (int j = 0; j < n; j++) { std::swap(a(p, j), a(col, j)); std::swap(out(p, j), out(col, j)); }
        const double inv = 1.0 / a(col, col);
        for (int j = 0; j < n; j++) { a(col, j) *= inv; out(col, j) *= inv; }
        for (int i = 0; i < n; i++) if (i != col) { const double f = a(i, col); if (f == 0) continue; for (int j = 0; j < n; j++) { a(i, j) -= f * a(col, j); out(i, j) -= f * out(col, j); } }
    }
    return true;
}
// Cholesky solve H x = b; false when H is not positive definite (CHOLMOD_NOT_POSDEF in the reference).
inline bool cholesky_solve(const Mat& H, const std::vector<double>& b, std::vector<double>& x) {
    const int n = H.r; Mat L(n, n);
    for (int j = 0; j < n; j++) {
        double d = H(j, j); for (int k = 0; k < j; k++) d -= L(j, k) * L(j, k);
        if (!(d > 0) || !std::isfinite(d)) return false;
        L(j, j) = std::sqrt(d);
        for (int i = j + 1; i < n; i++) { double s = H(i, j); for (int k = 0; k < j; k++) s -= L(i, k) * L(j, k); L(i, j) = s / L(j, j); }
    }
    x.assign(n, 0.0); std::vector<double> y(n);
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L(i, k) * y[k]; y[i] = s / L(i, i); }
    for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < n; k++) s -= L(k, i) * x[k]; x[i] = s / L(i, i); }
    return true;
}

} // namespace ora

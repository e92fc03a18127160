// tests/cpp/shim_tracking_test.cpp — compile / link / run test of viorb_amd/shim/viorb_tracking_shim.h, the templates a maintainer calls
// from Optimizer::PoseOptimization(Frame*, Frame* | KeyFrame*, ...). The reference's Frame / KeyFrame / MapPoint / NavState /
// IMUPreintegrator and Eigen / Sophus are absent from this image, so minimal stand-ins WITH THE REFERENCE'S MEMBER NAMES (the ones
// the shim touches: include/Frame.h, include/IMU/NavState.h, include/IMU/IMUPreintegrator.h) are defined here — test scaffolding only.
//   shim_tracking_test                      no device needed: runs both overloads on a tiny problem; without a device both must THROW
//                                           (viorb_shim::check surfaces viorb_last_error(); a GPU failure is never "0 inliers")
//   shim_tracking_test problem.bin out.bin  reads a problem written by tests/test_gpu_shims.py, writes what the shim stored in the frame
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "cv_standin.h"
#include "viorb_tracking_shim.h"

namespace standin {
struct Vec3 { double v[3]; Vec3(double x = 0, double y = 0, double z = 0) { v[0] = x; v[1] = y; v[2] = z; } double operator[](int i) const { return v[i]; } };
struct Quat { double w_, x_, y_, z_; Quat(double w = 1, double x = 0, double y = 0, double z = 0) : w_(w), x_(x), y_(y), z_(z) {}
              double x() const { return x_; } double y() const { return y_; } double z() const { return z_; } double w() const { return w_; } };
struct SO3 { Quat q; SO3() {} explicit SO3(const Quat& q_) : q(q_) {} const Quat& unit_quaternion() const { return q; } };
template <int R, int C> struct Mat { double m[R][C]; Mat() { for (auto& r : m) for (double& x : r) x = 0; } double& operator()(int r, int c) { return m[r][c]; } double operator()(int r, int c) const { return m[r][c]; } };
struct NavState {                                   // include/IMU/NavState.h:17-60
    Vec3 P, V, bg, ba, dbg, dba; SO3 R;
    Vec3 Get_P() const { return P; } Vec3 Get_V() const { return V; } SO3 Get_R() const { return R; }
    Vec3 Get_BiasGyr() const { return bg; } Vec3 Get_BiasAcc() const { return ba; } Vec3 Get_dBias_Gyr() const { return dbg; } Vec3 Get_dBias_Acc() const { return dba; }
    void Set_Pos(const Vec3& x) { P = x; } void Set_Vel(const Vec3& x) { V = x; } void Set_Rot(const SO3& x) { R = x; }
    void Set_BiasGyr(const Vec3& x) { bg = x; } void Set_BiasAcc(const Vec3& x) { ba = x; } void Set_DeltaBiasGyr(const Vec3& x) { dbg = x; } void Set_DeltaBiasAcc(const Vec3& x) { dba = x; }
};
struct Preint {                                     // include/IMU/IMUPreintegrator.h:40-75
    Vec3 dP, dV; Mat<3, 3> dR, JPg, JPa, JVg, JVa, JRg; Mat<9, 9> cov; double dt = 0;
    Vec3 getDeltaP() const { return dP; } Vec3 getDeltaV() const { return dV; } const Mat<3, 3>& getDeltaR() const { return dR; }
    const Mat<3, 3>& getJPBiasg() const { return JPg; } const Mat<3, 3>& getJPBiasa() const { return JPa; } const Mat<3, 3>& getJVBiasg() const { return JVg; }
    const Mat<3, 3>& getJVBiasa() const { return JVa; } const Mat<3, 3>& getJRBiasg() const { return JRg; } const Mat<9, 9>& getCovPVPhi() const { return cov; }
    double getDeltaTime() const { return dt; }
};
struct MapPoint { cv::Mat Pw; cv::Mat GetWorldPos() const { return Pw; } };
struct Frame {                                      // include/Frame.h: the members Optimizer::PoseOptimization reads and writes
    float fx = 0, fy = 0, cx = 0, cy = 0; int N = 0;
    std::vector<MapPoint*> mvpMapPoints; std::vector<float> mvuRight; std::vector<cv::KeyPoint> mvKeysUn; std::vector<float> mvInvLevelSigma2;
    std::vector<bool> mvbOutlier; NavState ns, mNavStatePrior; Mat<12, 12> mMargCovInv; int pose_updates = 0;
    const NavState& GetNavState() const { return ns; } void SetNavState(const NavState& x) { ns = x; } void UpdatePoseFromNS(const cv::Mat&) { pose_updates++; }
};
typedef Frame KeyFrame;                             // the KeyFrame overload only reads GetNavState()
}

static void unpack(const double* o, standin::NavState& ns) { viorb_shim::unpack_navstate<standin::NavState, standin::Vec3, standin::Quat, standin::SO3>(o, ns); }
static void fill_preint(const double* o, standin::Preint& M) {
    M.dP = standin::Vec3(o[0], o[1], o[2]); M.dV = standin::Vec3(o[3], o[4], o[5]);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        M.dR(r, c) = o[6 + 3 * r + c]; M.JPg(r, c) = o[15 + 3 * r + c]; M.JPa(r, c) = o[24 + 3 * r + c];
        M.JVg(r, c) = o[33 + 3 * r + c]; M.JVa(r, c) = o[42 + 3 * r + c]; M.JRg(r, c) = o[51 + 3 * r + c];
    }
    for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) M.cov(r, c) = o[60 + 9 * r + c];
    M.dt = o[141];
}
static void fill_frame(standin::Frame& F, std::vector<standin::MapPoint>& pts, const double* cam, const double* ns22, const double* obs, int n) {
    F.fx = (float)cam[0]; F.fy = (float)cam[1]; F.cx = (float)cam[2]; F.cy = (float)cam[3];
    unpack(ns22, F.ns);
    F.N = n; pts.resize(n); F.mvpMapPoints.assign(n, nullptr); F.mvuRight.assign(n, -1.f); F.mvKeysUn.resize(n); F.mvbOutlier.assign(n, false);
    F.mvInvLevelSigma2.resize(8);
    for (int l = 0; l < 8; l++) F.mvInvLevelSigma2[l] = (float)(1.0 / std::pow((double)(float)std::pow(1.2f, l), 2));
    for (int i = 0; i < n; i++) {
        pts[i].Pw.fbuf.assign(3, 0.f);
        for (int c = 0; c < 3; c++) pts[i].Pw.at<float>(c) = (float)obs[6 * i + c];
        F.mvpMapPoints[i] = &pts[i];
        // the octave is recovered from the inverse sigma^2 the problem file carries (the shim reads mvInvLevelSigma2[kp.octave])
        int oct = 0; double best = 1e30;
        for (int l = 0; l < 8; l++) { const double d = std::fabs((double)F.mvInvLevelSigma2[l] - obs[6 * i + 5]); if (d < best) { best = d; oct = l; } }
        F.mvKeysUn[i] = cv::KeyPoint((float)obs[6 * i + 3], (float)obs[6 * i + 4], 31.f, -1.f, 0.f, oct, -1);
    }
}

int main(int argc, char** argv) {
    using namespace standin;
    std::vector<double> in;
    if (argc >= 2) {
        FILE* f = fopen(argv[1], "rb"); if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
        fseek(f, 0, SEEK_END); const long bytes = ftell(f); fseek(f, 0, SEEK_SET);
        in.resize(bytes / sizeof(double)); if (fread(in.data(), sizeof(double), in.size(), f) != in.size()) return 2; fclose(f);
    } else {                                        // a tiny well-formed problem: 4 points in front of an identity pose
        in.assign(22 * 3 + 144 + 142 + 3 + 16 + 2, 0.0);
        for (int k = 0; k < 3; k++) in[22 * k + 9] = 1.0;                                 // unit quaternions
        double* pre = &in[66 + 144]; pre[6] = pre[10] = pre[14] = 1.0; for (int d = 0; d < 9; d++) pre[60 + 10 * d] = 1e-4; pre[141] = 0.05;
        double* cam = &in[66 + 144 + 142 + 3]; cam[0] = cam[1] = 450; cam[2] = 376; cam[3] = 240; cam[4] = cam[8] = cam[12] = 1.0;
        in[66 + 144 + 142 + 3 + 16] = 4; 
        const double P[4][3] = {{0.5, 0.2, 4}, {-0.4, 0.3, 5}, {0.1, -0.5, 6}, {-0.2, -0.1, 3}};
        for (int i = 0; i < 4; i++) { const double o[6] = {P[i][0], P[i][1], P[i][2], 450 * P[i][0] / P[i][2] + 376, 450 * P[i][1] / P[i][2] + 240, 1.0}; in.insert(in.end() - 1, o, o + 6); }
        in.back() = 0;                                                                   // no last-frame observations
    }
    const double* cur = &in[0]; const double* last = &in[22]; const double* prior = &in[44]; const double* mci = &in[66];
    const double* pre = &in[66 + 144]; const double* gw = pre + 142; const double* cam = gw + 3;
    const int nc = (int)cam[16]; const double* oc = cam + 17; const int nl = (int)oc[6 * nc]; const double* ol = oc + 6 * nc + 1;
    Mat<4, 4> Tbc; for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Tbc(r, c) = cam[4 + 3 * r + c]; Tbc(r, 3) = cam[13 + r]; } Tbc(3, 3) = 1;
    Preint M; fill_preint(pre, M);
    cv::Mat MatTbc;
    std::vector<double> out;
    for (int variant = 1; variant >= 0; variant--) {
        Frame F, L; std::vector<MapPoint> pc, pl;
        fill_frame(F, pc, cam, cur, oc, nc); fill_frame(L, pl, cam, last, ol, nl);
        unpack(prior, L.mNavStatePrior);
        for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) L.mMargCovInv(r, c) = mci[12 * r + c];
        int inl = 0;
        try {
            inl = variant ? viorb_shim::pose_optimization_frame<Vec3, Quat, SO3>(&F, &L, M, gw, Tbc, MatTbc, true)
                          : viorb_shim::pose_optimization_keyframe<Vec3, Quat, SO3>(&F, static_cast<KeyFrame*>(&L), M, gw, Tbc, MatTbc, true);
            if (viorb_device_count() < 1) { printf("FAIL: no device, but the shim returned %d instead of throwing\n", inl); return 1; }
        } catch (const std::runtime_error& e) {
            if (viorb_device_count() >= 1) { printf("FAIL: %s\n", e.what()); return 1; }
            if (F.pose_updates != 0) { printf("FAIL: frame touched before the error\n"); return 1; }
            if (variant == 0) { printf("OK (no device: both overloads threw \"%s\", frame untouched)\n", e.what()); return 0; }
            continue;
        }
        double ns[22]; viorb_shim::pack_navstate(F.GetNavState(), ns);
        out.push_back((double)inl); out.push_back((double)F.pose_updates);
        out.insert(out.end(), ns, ns + 22);
        for (int i = 0; i < nc; i++) out.push_back(F.mvbOutlier[i] ? 1.0 : 0.0);
        for (int i = 0; i < nl; i++) out.push_back(variant && L.mvbOutlier[i] ? 1.0 : 0.0);
        for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) out.push_back(F.mMargCovInv(r, c));
    }
    if (argc >= 3) { FILE* f = fopen(argv[2], "wb"); if (!f) return 2; fwrite(out.data(), sizeof(double), out.size(), f); fclose(f); }
    printf("OK inliers %d / %d (device count %d)\n", (int)out[0], (int)out[2 + 22 + nc + nl + 144], viorb_device_count());
    return 0;
}

// oracle/vio.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the visual-inertial part of the hot path: IMU pre-integration, NavState, the
// five VI g2o edge types, g2o's Levenberg-Marquardt with Huber kernels and the two
// Optimizer::PoseOptimization overloads with NavState edges (reference files cited per function in
// vio.cpp). Eigen3 / CHOLMOD / g2o's graph machinery are replaced by small dense FP64 code; the
// arithmetic that defines results (residuals, Jacobians, weighting, LM schedule, stop rule, chi2
// gating) is restated step for step. PARITY UNPINNED: the reference holds no test or golden vector
// for this path; pinned by definitional checks in tests/test_oracle_vio.py (numerical Jacobians,
// scipy rotations, closed-form cases).
#pragma once
#include "vio_math.h"
#include <vector>
#include <cstdint>

namespace ora {

// reference src/IMU/imudata.cpp:31-41 (the un-commented constants)
struct ImuNoise {
    static constexpr double gyrBiasRw2 = 2e-5 * 2e-5;
    static constexpr double accBiasRw2 = 5e-3 * 5e-3;
    static constexpr double gyrMeasCov = 2.0e-3 * 2.0e-3 * 200;     // x Identity
    static constexpr double accMeasCov = 8.0e-3 * 8.0e-3 * 200;
};

struct ImuSample { double g[3], a[3], t; };

// reference src/IMU/IMUPreintegrator.{h,cpp}
struct Preint {
    V3 dP, dV; M3 dR = M3::identity();
    M3 JPg, JPa, JVg, JVa, JRg;
    Mat cov = Mat(9, 9);
    double dt = 0;
    void reset();
    void update(V3 omega, V3 acc, double dt);          // IMUPreintegrator::update :86-153
};
// Frame::ComputeIMUPreIntSinceLastFrame (reference src/Frame.cc:41-86) /
// Tracking::GetIMUPreIntSinceLastKF (src/Tracking.cc:537-584): first sample also covers
// [t_last, t_imu0], the last sample is held until t_cur.
void preintegrate(const ImuSample* s, int n, V3 bg, V3 ba, double t_last, double t_cur, Preint& out);

// reference src/IMU/NavState.{h,cpp}
struct NavState {
    V3 P, V; SO3 R; V3 bg, ba, dbg, dba;
    void inc_small_pvr(const double* u9);              // IncSmallPVR :71-...
    void inc_small_bias(const double* u3);             // IncSmallBias (NOT_UPDATE_GYRO_BIAS: acc only)
};
// Converter::updateNS, reference src/Converter.cc:27-49
void update_ns(NavState& ns, const Preint& p, V3 gw);
// Tracking::PredictNavStateByIMU (src/Tracking.cc:348-410): Frame::SetInitialNavStateAndBias(last)
// (src/Frame.cc:117-125: bias += delta bias, delta = 0) followed by UpdateNavState.
NavState predict_navstate(const NavState& last, const Preint& p, V3 gw);

struct Camera { double fx, fy, cx, cy; M3 Rbc; V3 Pbc; };
// Frame::UpdatePoseFromNS (reference src/Frame.cc:88-105) in the float arithmetic of its cv::Mat
// expressions; pose12 = Rcw (row-major) then tcw. Same small-matrix gemm evaluation order as
// transform_point() in orb_matcher.cpp (parity unpinned vs OpenCV).
void pose_from_navstate_f32(const NavState& ns, const Camera& cam, float* pose12);

struct Observation { V3 Pw; double u, v, inv_sigma2; };

struct PoseOptResult {
    NavState ns;                    // optimised current-frame state
    NavState ns_last;               // optimised last-frame state (Frame/Frame variant; unchanged otherwise)
    std::vector<uint8_t> outlier_cur, outlier_last;
    int n_inliers = 0;              // return value of the reference function
    double final_chi2 = 0;          // activeRobustChi2 after the last round
    Mat marg_cov_inv;               // 12x12 mMargCovInv when requested
    int lm_iterations = 0;          // total LM outer iterations over the 4 rounds
    std::vector<double> chi2_trace; // robust chi2 after every outer LM iteration
};

// Optimizer::PoseOptimization(Frame*, KeyFrame*, preint, gw, marg), reference src/Optimizer.cc:789-1112
PoseOptResult pose_opt_vi_kf(const NavState& cur, const NavState& last_kf, const Preint& preint, V3 gw,
                             const Camera& cam, const std::vector<Observation>& obs_cur, bool compute_marg);
// Optimizer::PoseOptimization(Frame*, Frame*, preint, gw, marg), reference src/Optimizer.cc:323-787
PoseOptResult pose_opt_vi_frame(const NavState& cur, const NavState& last, const NavState& last_prior,
                                const Mat& last_marg_cov_inv, const Preint& preint, V3 gw, const Camera& cam,
                                const std::vector<Observation>& obs_cur, const std::vector<Observation>& obs_last,
                                bool compute_marg);

// Diagnostics of the last pose_opt_vi_* call on this thread (tests): rounds whose optimize() ended on a rejected trial, and inlier edges whose
// verdict by g2o's stored (stale) error differs from the verdict at the restored estimate.
void pose_opt_diagnostics(int* rejected_rounds, int* stale_verdicts);

// ---- vision-only pose optimisation, Optimizer::PoseOptimization(Frame*), reference src/Optimizer.cc:3749-3978,
// edges Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:266-364, SE3Quat::exp se3quat.h:223-257.
struct Se3Obs { V3 Xw; double u, v, ur; double inv_sigma2; };     // ur < 0: monocular edge, else stereo edge
struct Se3Result {
    float pose12[12];                   // Converter::toCvMat(SE3Quat): Rcw row-major + tcw, float
    std::vector<uint8_t> outlier;
    int n_inliers = 0; double final_chi2 = 0; int lm_iterations = 0;
};
Se3Result pose_opt_se3(const float* pose12, double fx, double fy, double cx, double cy, double bf,
                       const std::vector<Se3Obs>& obs);

// Residual / Jacobian blocks of the individual edges, exposed for the definitional tests.
void edge_pvr_error(const NavState& i, const NavState& j, const NavState& bias_i, const Preint& M, V3 gw, double* e9);
void edge_pvr_jacobians(const NavState& i, const NavState& j, const NavState& bias_i, const Preint& M, V3 gw,
                        const double* e9, Mat& Ji, Mat& Jj, Mat& Jb);
void edge_proj_error(const NavState& ns, const Camera& cam, const Observation& o, double* e2);
void edge_proj_jacobian(const NavState& ns, const Camera& cam, const Observation& o, Mat& J);   // 2x9
void edge_prior_error(const NavState& pvr, const NavState& bias, const NavState& prior, double* e12);
void edge_prior_jacobians(const NavState& pvr, const double* e12, Mat& Jpvr, Mat& Jbias);

} // namespace ora

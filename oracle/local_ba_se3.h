// oracle/local_ba_se3.h — TEST INFRASTRUCTURE ONLY. CPU restatement of the vision-only Optimizer::LocalBundleAdjustment
// (reference src/Optimizer.cc:3980-4311): free VertexSE3Expmap key frames (left-multiplicative exp update,
// Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:73-76), fixed key frames, marginalised VertexSBAPointXYZ points,
// EdgeSE3ProjectXYZ (mono, 2-D) and EdgeStereoSE3ProjectXYZ (stereo, 3-D, float reciprocal depth) edges
// (types_six_dof_expmap.cpp:66-250), BlockSolver_6_3 = Levenberg with the Schur complement of the point block.
// optimize(5) -> chi2 > 5.991 / 7.815 or non-positive depth to level 1, robust kernels dropped -> optimize(10).
// PARITY UNPINNED (no reference fixture); pinned by tests/test_oracle_local_ba.py (dense scipy re-optimisation).
#pragma once
#include "vio.h"
namespace ora {
struct Se3Pose { Quat r; V3 t; };                                  // g2o::SE3Quat (Tcw)
struct BaSe3Edge { int point, kf; double u, v, ur, inv_sigma2; };   // ur < 0: monocular observation
struct BaSe3Problem {
    std::vector<Se3Pose> kfs; int n_local = 0;                       // free key frames first, then fixed ones (KF 0 counts as fixed)
    std::vector<V3> points; std::vector<BaSe3Edge> edges;           // edges grouped by point
    double fx, fy, cx, cy, bf;
};
struct BaSe3Result {
    std::vector<Se3Pose> kfs; std::vector<V3> points; std::vector<uint8_t> erase;
    double chi2_after_first = 0, chi2_final = 0; int its_first = 0, its_second = 0;
};
BaSe3Result local_ba_se3(const BaSe3Problem& P, const volatile int* stop);
}

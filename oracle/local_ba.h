// oracle/local_ba.h — TEST INFRASTRUCTURE ONLY. CPU restatement of Optimizer::LocalBundleAdjustmentNavState
// (reference src/Optimizer.cc:1690-2241) on a flat graph description: W local key frames with free PVR(9)+Bias(3)
// vertices, fixed key frames (PVR only; the previous key frame of the window also fixes the IMU chain), marginalised
// 3-D points, one IMU factor + one bias factor per local key frame (information NOT inflated, :1901-1919), one
// EdgeNavStatePVRPointXYZ per observation (src/IMU/g2otypes.h:129-203, g2otypes.cpp:299-354). g2o's LM with Schur
// complement (Thirdparty/g2o/g2o/core/block_solver.hpp:367-486) on a dense reduced system.
// PARITY UNPINNED (no reference fixture); pinned by tests/test_oracle_local_ba.py.
#pragma once
#include "vio.h"
namespace ora {
struct BaEdge { int point, kf; double u, v, inv_sigma2; };         // kf indexes ALL key frames: [0, n_local) free, then fixed
struct BaProblem {
    std::vector<NavState> kfs;          // local key frames first (window order), then fixed ones
    int n_local = 0;
    int prev_kf = -1;                   // index (>= n_local) of the fixed previous key frame of kfs[0], or -1
    std::vector<Preint> preint;         // preint[i]: IMU between the predecessor of local KF i and local KF i
    std::vector<V3> points;
    std::vector<BaEdge> edges;          // grouped by point, in the reference's construction order
    V3 gw; Camera cam;
};
struct BaResult {
    std::vector<NavState> kfs;          // optimised local key frames (dBias updated)
    std::vector<V3> points;
    std::vector<uint8_t> erase;         // per edge: chi2 > 5.991 or negative depth after the second optimisation
    double chi2_after_first = 0, chi2_final = 0;
    int its_first = 0, its_second = 0;
    std::vector<double> trace;
};
// stop: checked where g2o polls terminate() (before each optimize iteration and each LM trial); may be null.
BaResult local_ba_navstate(const BaProblem& P, const volatile int* stop);
}

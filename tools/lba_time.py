"""Wall-clock of one LocalBundleAdjustmentNavState on the GPU path vs the oracle (W=20, 2000 points)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as o
from viorb_amd.synth import make_local_ba_problem
from viorb_amd import LocalBundleAdjustmentNavState
p = make_local_ba_problem(3, W=20, n_points=2000)
pre = []
for i, (imu, t0, t1) in enumerate(p["imu"]):
    j = i - 1 if i > 0 else p["prev_kf"]
    pre.append(o.preintegrate(imu, p["kfs"][j][10:13], p["kfs"][j][13:16], t0, t1))
pre = np.stack(pre)
a = (p["kfs"], p["n_local"], p["prev_kf"], pre, p["points"], p["edge_idx"], p["edge_obs"], p["gw"], p["cam"])
for name, f in (("oracle", o.local_ba), ("gpu", LocalBundleAdjustmentNavState)):
    f(*a)
    t0 = time.perf_counter()
    for _ in range(5):
        r = f(*a)
    print(name, "ms/solve %.2f" % ((time.perf_counter() - t0) / 5 * 1e3), r["its_first"], r["its_second"], r["chi2_final"], flush=True)

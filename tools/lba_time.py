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

# throughput with several windows in flight (viorb_local_ba_navstate_batch: one host thread, one HIP stream per window)
from viorb_amd import LocalBundleAdjustmentNavStateBatch
q = dict(kfs=a[0], n_local=a[1], prev_kf=a[2], preint=a[3], points=a[4], edge_idx=a[5], edge_obs=a[6], gw=a[7], cam=a[8])
for nwin, fl in ((8, 1), (8, 2), (16, 4), (32, 8), (64, 16), (64, 32)):
    LocalBundleAdjustmentNavStateBatch([q] * min(nwin, fl), max_in_flight=fl)
    t0 = time.perf_counter()
    LocalBundleAdjustmentNavStateBatch([q] * nwin, max_in_flight=fl)
    dt = time.perf_counter() - t0
    print("batch of %d windows, %d in flight: %.2f ms per window, %.0f windows/s" % (nwin, fl, dt / nwin * 1e3, nwin / dt), flush=True)

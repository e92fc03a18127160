import sys, os
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
r = LocalBundleAdjustmentNavState(*a)
print(r["its_first"], r["its_second"])

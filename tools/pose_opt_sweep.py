"""Randomised parity sweep of the VI pose solver against the oracle (both overloads, 60 seeds each, 200-700 points): discrete results
(inliers, outlier flags, LM iteration counts) must be identical, cost within 1e-5, state within 1e-7. Dev aid after numerical changes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import viorb_amd
from oracle import binding as oracle
from viorb_amd.synth import make_vio_problem
bad = 0; n = 0
for variant in (0, 1):
    for seed in range(100, 160):
        p = make_vio_problem(seed, n_points=200 + 13 * (seed % 40))
        last = p["ns_last"]
        pre = oracle.preintegrate(p["imu"], last[10:13], last[13:16], p["t_last"], p["t_cur"])
        cur0 = oracle.update_ns(last, pre, p["gw"])
        if variant == 0:
            o = oracle.pose_opt_vi_kf(cur0, last, pre, p["gw"], p["cam"], p["obs_cur"], marg=True)
            g = viorb_amd.PoseOptimization(cur0, last, pre, p["gw"], p["cam"], p["obs_cur"], last_is_keyframe=True, bComputeMarg=True)
        else:
            o = oracle.pose_opt_vi_frame(cur0, last, p["prior"], p["marg_cov_inv"], pre, p["gw"], p["cam"], p["obs_cur"], p["obs_last"], marg=True)
            g = viorb_amd.PoseOptimization(cur0, last, pre, p["gw"], p["cam"], p["obs_cur"], p["obs_last"], p["prior"], p["marg_cov_inv"], last_is_keyframe=False, bComputeMarg=True)
        n += 1
        ok = (g["n_inliers"] == o["n_inliers"] and g["lm_iterations"] == o["lm_iterations"] and np.array_equal(g["outlier_cur"], o["outlier_cur"])
              and abs(g["final_chi2"] - o["final_chi2"]) <= 1e-5 * abs(o["final_chi2"]) and np.allclose(g["ns"], o["ns"], rtol=0, atol=1e-7))
        if not ok:
            bad += 1; print("MISMATCH variant", variant, "seed", seed, g["n_inliers"], o["n_inliers"], g["lm_iterations"], o["lm_iterations"], g["final_chi2"], o["final_chi2"])
print("checked", n, "problems, mismatches", bad)

"""Wall-clock of the single-stream host-buffer drop-ins (what a Tracking thread pays per call through the shims): ORBextractor::operator(),
SearchByProjection(Frame, Frame), PoseOptimization(Frame, Frame), SearchLocalPoints. Dev aid."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import viorb_amd
from oracle import binding as ora
from viorb_amd.synth import make_vi_stream, make_vio_problem, backproject_to_plane, cam_pose_from_navstate
s = make_vi_stream(1, 2)
ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7)
k0, d0 = ex(s["frames"][0]); k1, d1 = ex(s["frames"][1])
def timeit(name, f, n=50):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n
    print("%-40s %.3f ms per call (%.0f calls/s)" % (name, dt * 1e3, 1 / dt))
timeit("viorb_extract 752x480", lambda: ex(s["frames"][1]))
cam = s["cam"]; sf = ex.tables()["scale"]
Pw = backproject_to_plane(np.stack([k0["x"], k0["y"]], 1).astype(np.float64), s["ns_true"][0], cam).astype(np.float32)
Rcw, tcw = cam_pose_from_navstate(s["ns_true"][1], cam); pose = np.concatenate([Rcw.ravel(), tcw]).astype(np.float32)
flags = np.full(len(k0), 5, np.uint8)
m = viorb_amd.ORBmatcher(0.9, True)
timeit("viorb_search_by_projection_frame", lambda: m.SearchByProjection(k1, d1, (0, 752, 0, 480), pose, cam[:4], sf, k0, flags, Pw, d0, 15.0))
p = make_vio_problem(3, n_points=700)
last = p["ns_last"]; pre = ora.preintegrate(p["imu"], last[10:13], last[13:16], p["t_last"], p["t_cur"]); cur0 = ora.update_ns(last, pre, p["gw"])
timeit("viorb_pose_opt_vi (Frame, Frame)", lambda: viorb_amd.PoseOptimization(cur0, last, pre, p["gw"], p["cam"], p["obs_cur"], p["obs_last"], p["prior"], p["marg_cov_inv"], last_is_keyframe=False, bComputeMarg=True))
timeit("viorb_preintegrate", lambda: viorb_amd.preintegrate(p["imu"], last[10:13], last[13:16], p["t_last"], p["t_cur"]))

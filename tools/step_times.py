"""Per-step completion times of the default bench sequence over a long run (is the step time stationary?). Dev aid."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from viorb_amd.tracker import BatchedTracker
S = 256; K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
streams = bench.generate_streams(list(range(1000, 1000 + S)))
dev = torch.device("cuda", 0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
frames = up(np.stack([s["frames"] for s in streams], 1)); imu = up(np.stack([s["imu"] for s in streams], 1))
t_frames = up(np.stack([s["t"] for s in streams], 1)); pose_true = up(np.stack([s["pose_true"] for s in streams], 1)); ns_true = up(np.stack([s["ns_true"] for s in streams], 1))
period = up(np.array([s["period"] for s in streams])); zeros = torch.zeros(S, dtype=torch.float64, device=dev)
mci0 = up(np.stack([np.eye(12).ravel() * 1e3] * S))
tr = BatchedTracker(streams[0]["cam"], streams[0]["gw"], S, track_local_map=True)
tr.bootstrap(frames[0], pose_true[0], t_frames[0], ns_true[0], mci0)
N = frames.shape[0]
evs = []; its = []
def step(k):
    j = k % N
    if j == 0: tr.step(frames[0], imu[0], period, pose_true[0], t_next_last=zeros, chain_estimate=False, true_ns=ns_true[0], marg_reset=mci0)
    else: tr.step(frames[j], imu[j], t_frames[j], pose_true[j])
    e = torch.cuda.Event(enable_timing=True); e.record(tr.s_tr); evs.append(e)
    if k % 25 == 0:
        torch.cuda.synchronize(); its.append((k, float(tr.info[:, 2].mean()), float(tr.info2[:, 2].mean()), float(tr.info2[:, 0].mean()), float(tr.info2[:, 1].mean())))
for k in range(1, K + 1): step(k)
torch.cuda.synchronize()
dt = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)])
for a in range(0, len(dt), 25): print("steps %3d-%3d: mean %.3f ms  min %.3f  max %.3f" % (a, min(a + 25, len(dt)), dt[a:a + 25].mean(), dt[a:a + 25].min(), dt[a:a + 25].max()))
for r in its: print("step %3d: mean LM iterations first solve %.1f, second solve %.1f; inliers %.0f; chi2 %.1f" % r)

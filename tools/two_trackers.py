"""Experiment: G trackers of S / G streams each, driven by G host threads and started a fraction of a step apart, against one tracker of S
streams (does a phase offset between pipelines smooth the contention between the pose solver and the extractor?). Dev aid.
usage: two_trackers.py [S] [G] [steps]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, viorb_amd
import bench
from viorb_amd.tracker import NativeTracker
from viorb_amd.synth import EUROC_DIST
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 80
distinct = min(S // G, 128)
streams = bench.generate_streams(list(range(1000, 1000 + distinct)), 752, 480, None, EUROC_DIST)
dev = torch.device("cuda", 0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
Sg = S // G
idx = torch.arange(Sg, device=dev) % distinct
rep = lambda a: up(a)[:, idx].contiguous()
frames = rep(np.stack([s["frames"] for s in streams], 1)); imu = rep(np.stack([s["imu"] for s in streams], 1))
t_frames = rep(np.stack([s["t"] for s in streams], 1)); pose_true = rep(np.stack([s["pose_true"] for s in streams], 1)); ns_true = rep(np.stack([s["ns_true"] for s in streams], 1))
t_period = up(np.array([s["period"] for s in streams]))[idx].contiguous()
zeros_t = torch.zeros(Sg, dtype=torch.float64, device=dev); ones_u8 = torch.ones(Sg, dtype=torch.uint8, device=dev)
mci0 = up(np.stack([np.eye(12).ravel() * 1e3] * Sg))
cam, gw = streams[0]["cam"], streams[0]["gw"]
lens = EUROC_DIST
NF = frames.shape[0]
trs = []
for g in range(G):
    tr = NativeTracker(cam, gw, Sg, 752, 480, 1000, th=15.0, device=0, compute_marg=True, track_local_map=True, dist_coef=lens)
    tr.bootstrap(frames[0], pose_true[0], t_frames[0], ns_true[0], mci0)
    trs.append(tr)
def step(tr, k):
    j = k % NF
    if j == 0:
        tr.step(frames[0], imu[0], t_period, pose_true[0], t_next_last=zeros_t, reset_ns=ns_true[0], reset_marg=mci0)
    elif j == 1 and k > 1:
        tr.step(frames[1], imu[1], t_frames[1], pose_true[1], map_updated=ones_u8)
    else:
        tr.step(frames[j], imu[j], t_frames[j], pose_true[j])
def worker(g, n, delay):
    time.sleep(delay)
    k = 1
    for _ in range(n):
        step(trs[g], k); k += 1
    trs[g].sync()
for phase in (0.0, 0.5):
    for n in (16, STEPS):                      # warm-up, then timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # one step of S / G streams takes ~ (5.7 ms / G): the offset is `phase` of that
        th = [threading.Thread(target=worker, args=(g, n, g * phase * 0.0057 / G)) for g in range(G)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print("S %d in %d trackers, start offset %.1f step: %.1f k frames/s (%.3f ms per %d frames)" % (S, G, phase, S * STEPS / el / 1e3, el / STEPS * 1e3, S), flush=True)

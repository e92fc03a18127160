"""Throughput of viorb_local_ba_se3_batch (vision-only LocalBundleAdjustment windows, W = 8 key frames, 600 points, half stereo):
lock-step batch by default, VIORB_LBA_STREAMS=1 = the per-stream driver."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viorb_amd.synth import make_local_ba_se3_problem
import viorb_amd
from viorb_amd import frontend as F
p = make_local_ba_se3_problem(4)
q = {k: p[k] for k in ("kfs", "n_local", "points", "edge_idx", "edge_obs", "intr5")}
for nwin, fl in ((8, 8), (64, 32), (256, 32)):
    F.LocalBundleAdjustmentBatch([q] * min(nwin, 8), max_in_flight=fl)
    t0 = time.perf_counter()
    F.LocalBundleAdjustmentBatch([q] * nwin, max_in_flight=fl)
    dt = time.perf_counter() - t0
    print("%d windows: %.2f ms per window, %.0f windows/s" % (nwin, dt / nwin * 1e3, nwin / dt), flush=True)

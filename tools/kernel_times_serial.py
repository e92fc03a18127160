"""Per-kernel times of one 256-stream step with the two-stream overlap switched off (every kernel alone on the GPU): the
reference point for how much each kernel is inflated by running beside the other stream in bench.py. Dev aid."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, viorb_amd
import bench
from viorb_amd.tracker import BatchedTracker
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
streams = bench.generate_streams(list(range(1000, 1000 + S)))
dev = torch.device("cuda", 0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
frames = up(np.stack([s["frames"] for s in streams], 1)); imu = up(np.stack([s["imu"] for s in streams], 1))
t_frames = up(np.stack([s["t"] for s in streams], 1)); pose_true = up(np.stack([s["pose_true"] for s in streams], 1)); ns_true = up(np.stack([s["ns_true"] for s in streams], 1))
mci0 = up(np.stack([np.eye(12).ravel() * 1e3] * S))
tr = BatchedTracker(streams[0]["cam"], streams[0]["gw"], S, overlap=False, track_local_map=True)
tr.bootstrap(frames[0], pose_true[0], t_frames[0], ns_true[0], mci0)
L = viorb_amd.lib()
for j in (1, 2, 3):
    tr.step(frames[j], imu[j], t_frames[j], pose_true[j])
torch.cuda.synchronize(); L.viorb_profile_reset(); L.viorb_profile_enable(1)
K = 4
for j in (4, 5, 6, 7):
    tr.step(frames[j], imu[j], t_frames[j], pose_true[j])
torch.cuda.synchronize(); L.viorb_profile_enable(0)
names = C.create_string_buffer(4096); ms = (C.c_double * 64)(); calls = (C.c_int * 64)(); k = C.c_int()
L.viorb_profile_read(names, 4096, ms, calls, 64, C.byref(k))
tot = 0
for i, nm in enumerate(names.value.decode().split("\n")[:k.value]):
    print("%-26s %7.3f ms per step" % (nm, ms[i] / K)); tot += ms[i] / K
print("sum %.3f ms per %d-stream step (serial)" % (tot, S))

"""Throughput of the stereo front-end of BASELINE.json configs[2] (KITTI-shaped 1241x376 pairs, 2000 features per image): ORB
extraction of left and right images in one batched handle + Frame::ComputeStereoMatches, P pairs per launch. Dev aid / evidence for
DESIGN.md; bench.py (the contract) measures configs[1]."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import viorb_amd
from viorb_amd.capi import lib, check, ptr
from viorb_amd.synth import make_stereo_pair, KITTI_K
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pairs = [make_stereo_pair(100 + s, 1241, 376)[:2] for s in range(min(P, 8))]
imgs = torch.from_numpy(np.stack([pairs[i % len(pairs)][0] for i in range(P)] + [pairs[i % len(pairs)][1] for i in range(P)])).cuda()
ex = viorb_amd.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=2 * P)
u = torch.zeros((P, ex.cap), dtype=torch.float32, device="cuda"); d = torch.zeros_like(u); n = torch.zeros(P, dtype=torch.int32, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def step():
    ex.extract_batch_device(imgs)
    check(lib().viorb_stereo_match_device(ex.h, 0, ex.h, P, P, KITTI_K["bf"], KITTI_K["fx"], ptr(u), ptr(d), ptr(n), st))
for _ in range(3): step()
torch.cuda.synchronize(); lib().viorb_profile_reset(); lib().viorb_profile_enable(1)
t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
lib().viorb_profile_enable(0)
names = C.create_string_buffer(4096); ms = (C.c_double * 64)(); calls = (C.c_int * 64)(); k = C.c_int()
lib().viorb_profile_read(names, 4096, ms, calls, 64, C.byref(k))
print("stereo pairs per launch %d: %.3f ms per launch, %.0f pairs/s; matched per pair %.0f of %d keypoints" % (P, dt * 1e3, P / dt, n.float().mean().item(), ex.cap))
print({nm: round(ms[i] / K, 3) for i, nm in enumerate(names.value.decode().split("\n")[:k.value])})

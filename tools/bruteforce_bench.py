"""Throughput of the brute-force Hamming tile matcher (k_match_bruteforce): B frame pairs of N x N descriptors per launch.
Prints Gpopcount32/s (Nq * Nc * 8 32-bit xor + popcount + add per pair of frames) against the chip's 32-bit
integer vector issue rate (256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz = 78.6 T lane-ops/s; one xor + one v_bcnt_u32_b32
(popcount with accumulate) per word = 39.3 T popcount32/s peak). Dev aid; run under rocprofv3 --kernel-trace --stats for the profile."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from viorb_amd.capi import lib, check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
q = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=dev, generator=g); c = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=dev, generator=g)
n = torch.full((B,), N, dtype=torch.int32, device=dev)
best, second, idx = (torch.zeros((B, N), dtype=torch.int32, device=dev) for _ in range(3))
vp = lambda t: C.c_void_p(t.data_ptr())
run = lambda: check(lib().viorb_match_bruteforce_device(vp(q), vp(n), N, vp(c), vp(n), N, B, vp(best), vp(second), vp(idx), None))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
pop = B * N * N * 8
PEAK = 256 * 4 * 32 * 2.4e9 / 2
print("k_match_bruteforce B=%d N=%d: %.3f ms per launch, %.1f Gpopcount32/s = %.3f of the %.1f T/s integer-issue peak, %.1f M frame pairs/s... %.0f pairs/s"
      % (B, N, ms, pop / ms / 1e6, pop / (ms * 1e-3) / PEAK, PEAK / 1e12, B / ms / 1e3, B / (ms * 1e-3)))

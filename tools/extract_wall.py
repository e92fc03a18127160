"""Wall-clock of the batched extractor alone (no per-kernel events): images/s and the extractor family's algorithmic HBM rate
B_ext = 4 P + K (709 + 961 + 60) per image against 8 TB/s.  python tools/extract_wall.py [B]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import viorb_amd
from viorb_amd.synth import make_image
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
base = [make_image(1000 + s, 752, 480) for s in range(8)]
imgs = torch.from_numpy(np.stack([base[i % 8] for i in range(B)])).cuda()
ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
for _ in range(3): ex.extract_batch_device(imgs)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): ex.extract_batch_device(imgs)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
P = 1117367; bext = 4 * P + 1000 * (709 + 961 + 60)
print("B=%d: %.3f ms per batch = %.3f ms per 256 images, %.0f images/s, %.2f TB/s algorithmic = %.1f %% of 8 TB/s" %
      (B, dt * 1e3, dt * 1e3 * 256 / B, B / dt, bext * B / dt / 1e12, bext * B / dt / 8e12 * 100))

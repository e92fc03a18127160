"""Quick extractor-only timing on the GPU (dev aid; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import viorb_amd
from viorb_amd.synth import make_image
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
imgs = torch.from_numpy(np.stack([make_image(s) for s in range(B)])).cuda()
ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
for _ in range(3):
    ex.extract_batch_device(imgs)
torch.cuda.synchronize()
K = 20
t = time.time()
for _ in range(K):
    ex.extract_batch_device(imgs)
torch.cuda.synchronize()
dt = (time.time() - t) / K
print("B=%d  %.3f ms/batch  %.1f fps  (%.1f us/frame)" % (B, dt * 1e3, B / dt, dt / B * 1e6))

"""Per-kernel times of the batched extractor alone (B EuRoC-shaped 752x480 frames per launch, 1000 features), each kernel timed with
HIP events. Dev aid: `python tools/extract_times.py [B] [library.so]` — the optional second argument loads another build of the
library (phase-elimination experiments on k_fast_cells)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    import viorb_amd.capi as capi
    capi.SO_PATH = os.path.abspath(sys.argv[2])
import numpy as np, torch
import viorb_amd
from viorb_amd.capi import lib
from viorb_amd.synth import make_image
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
base = [make_image(1000 + s, 752, 480) for s in range(8)]
imgs = torch.from_numpy(np.stack([base[i % 8] for i in range(B)])).cuda()
ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
for _ in range(2): ex.extract_batch_device(imgs)
torch.cuda.synchronize(); lib().viorb_profile_select(None); lib().viorb_profile_reset(); lib().viorb_profile_enable(1)
K = 5
for _ in range(K): ex.extract_batch_device(imgs)
torch.cuda.synchronize(); lib().viorb_profile_enable(0)
names = C.create_string_buffer(4096); ms = (C.c_double * 64)(); calls = (C.c_int * 64)(); k = C.c_int()
lib().viorb_profile_read(names, 4096, ms, calls, 64, C.byref(k))
print({nm: round(ms[i] / K, 3) for i, nm in enumerate(names.value.decode().split("\n")[:k.value])})

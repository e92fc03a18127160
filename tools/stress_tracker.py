"""Randomised run of the C++ batched tracker against the oracle twin (the comparison of tests/test_gpu_native_tracker.py on streams with
random seeds, with and without the EuRoC lens): `python tools/stress_tracker.py [n]` on a GPU box. A development aid."""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_native_tracker as tn
from viorb_amd.synth import EUROC_DIST
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(4242)
fails = 0
for i in range(N):
    seeds = [int(s) for s in rng.integers(300, 90000, 3)]
    dist = EUROC_DIST if i % 2 else None
    plan = dict(B=3, seeds=seeds, dist=dist, image=lambda b, j, st: st[b]["frames"][j], map_updated=lambda b, j: (j % 4 == 1) and j > 1,
                recent_reloc=lambda b, j: False, last_points=lambda b, j: None)
    try:
        seen = tn._run(752, 480, 1000, 7, plan)
        print("case", i, seeds, "lens" if dist else "pinhole", "states", sorted(seen))
    except AssertionError as e:
        fails += 1; print("FAIL", i, seeds, "lens" if dist else "pinhole", str(e)[:300])
    except Exception as e:
        fails += 1; print("ERROR", i, seeds, repr(e)[:300]); traceback.print_exc(limit=3)
print("cases", N, "failures", fails)

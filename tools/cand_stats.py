import sys, numpy as np
sys.path.insert(0, '/root/repo')
import viorb_amd
from viorb_amd.synth import make_periodic_stream
from viorb_amd import synth
import inspect
ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7)
dist = np.array([-0.28340811, 0.07395907, 0.00019359, 1.762e-05, 0.0])
rows = []
for seed in range(6):
    for d in (None, dist):
        s = make_periodic_stream(seed, 3, dist=d)
        for f in s["frames"][:2]:
            ex(f)
            rows.append([len(ex.debug_level_points(l)) for l in range(8)])
rows = np.array(rows)
print("per-level candidates: mean", rows.mean(0).round(0), "max", rows.max(0), "ratio to level 0 (mean)", (rows.mean(0) / rows.mean(0)[0]).round(2))
print("pinhole rows", rows[0::4].mean(0).round(0), "lens rows", rows[2::4].mean(0).round(0))

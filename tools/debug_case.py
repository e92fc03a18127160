"""Stage-by-stage comparison of one extractor configuration against the oracle (development aid): python tools/debug_case.py seed w h nf sf nl ini mn"""
import sys, numpy as np
sys.path.insert(0, ".")
import viorb_amd
from viorb_amd.synth import make_image
from oracle import binding as ora
seed, w, h, nf = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sf, nl, ini, mn = float(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
img = make_image(seed, w, h)
ex = viorb_amd.ORBextractor(nf, sf, nl, ini, mn); k, d = ex(img)
ox = ora.Extractor(nf, sf, nl, ini, mn); ok, od = ox(img)
print("keypoints", len(k), len(ok))
for l in range(nl):
    same_p = np.array_equal(ex.level(l), ox.level(l))
    oc = ox.level_keypoints(l, candidates=True); gc = ex.debug_level_points(l, kept=False)
    wantc = np.stack([oc["x"], oc["y"], oc["response"]], 1).astype(np.int32).reshape(-1, 3)
    okk = ox.level_keypoints(l); gk = ex.debug_level_points(l, kept=True)
    wantk = np.stack([okk["x"], okk["y"], okk["response"]], 1).astype(np.int32).reshape(-1, 3)
    print("level", l, ex.level(l).shape, "pyramid", same_p, "candidates", len(gc), len(wantc), np.array_equal(gc, wantc), "kept", len(gk), len(wantk), np.array_equal(gk, wantk))
    if not np.array_equal(gk, wantk) and np.array_equal(gc, wantc):
        sg = set(map(tuple, gk)); so = set(map(tuple, wantk))
        print("   only gpu:", sorted(sg - so)[:6], " only oracle:", sorted(so - sg)[:6], " same set:", sg == so)

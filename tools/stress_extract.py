"""Randomised differential run of the extractor against the oracle (sizes, feature counts, scale factors, level counts, thresholds, image kinds):
`python tools/stress_extract.py` on a GPU box; prints every mismatch. A development aid beside the fixed cases of tests/test_gpu_extractor.py."""
import sys, numpy as np
sys.path.insert(0, ".")
import viorb_amd
from viorb_amd.synth import make_image
from oracle import binding as ora
rng = np.random.default_rng(2026)
bad = 0; loud = 0
for it in range(72):
    w = int(rng.integers(96, 1400)); h = int(rng.integers(80, 800)); nf = int(rng.integers(50, 2500))
    sf = float(rng.choice([1.2, 1.1, 1.3, 1.5])); nl = int(rng.integers(1, 9)); ini = int(rng.choice([20, 12, 30])); mn = int(rng.choice([7, 5, 3]))
    if it >= 60: nf = int(rng.integers(2600, 6000)); nl = 8 if it % 2 else int(rng.integers(1, 4)); w = int(rng.integers(600, 1300)); h = int(rng.integers(400, 720))      # large quotas
    kind = it % 4
    img = make_image(7000 + it, w, h)
    if kind == 1: img = rng.integers(0, 256, (h, w), dtype=np.uint8)                       # noise
    if kind == 2: img = ((np.indices((h, w)).sum(0) // 3 % 2) * 200 + 20).astype(np.uint8)  # dense corners
    try:
        ex = viorb_amd.ORBextractor(nf, sf, nl, ini, mn); k, d = ex(img)
        ox = ora.Extractor(nf, sf, nl, ini, mn); ok, od = ox(img)
        same = k.shape == ok.shape and np.array_equal(k, ok) and np.array_equal(d, od)
    except viorb_amd.capi.ViorbError as e:                   # a refused configuration (documented limits): loud, not wrong
        loud += 1; print("refused", it, w, h, nf, sf, nl, str(e)[:110]); continue
    if not same:
        bad += 1; print("MISMATCH", it, w, h, nf, sf, nl, ini, mn, kind, len(k), len(ok))
print("cases 72, refused (loud error)", loud, "silent mismatches", bad)

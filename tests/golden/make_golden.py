#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle on seeded synthetic inputs.

The reference holds no golden vectors for this path and cannot be built or
imported here (C++ needing OpenCV 2.4 / Eigen3 / CHOLMOD, all absent), so these fixtures are outputs
of the build's own oracle: they pin the oracle against regressions and give the -m gpu tests
expected values that do not depend on recomputation. Inputs are regenerated from the seed
(viorb_amd.synth), only expected outputs are stored.
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import binding as ora
from viorb_amd.synth import make_image


def extractor_case(name, seed, w, h, nfeat):
    img = make_image(seed, w, h)
    ex = ora.Extractor(nfeat, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    cand_counts = np.array([len(ex.level_keypoints(l, candidates=True)) for l in range(8)], np.int32)
    lvl_sum = np.array([int(ex.level(l).astype(np.uint64).sum()) for l in range(8)], np.uint64)
    np.savez_compressed(os.path.join(HERE, name), seed=seed, w=w, h=h, nfeat=nfeat,
                        kps=kps, desc=desc, cand_counts=cand_counts, level_sums=lvl_sum)
    print(name, len(kps), cand_counts.tolist())


if __name__ == "__main__":
    extractor_case("extract_euroc_seed0.npz", 0, 752, 480, 1000)
    extractor_case("extract_euroc_seed1.npz", 1, 752, 480, 1000)
    extractor_case("extract_kitti_seed100.npz", 100, 1241, 376, 2000)
    extractor_case("extract_small_seed5.npz", 5, 160, 120, 300)

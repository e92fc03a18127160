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


def _preints(p):
    out = []
    for i, (imu, t0, t1) in enumerate(p["imu"]):
        j = i - 1 if i > 0 else p["prev_kf"]
        out.append(ora.preintegrate(imu, p["kfs"][j][10:13], p["kfs"][j][13:16], t0, t1))
    return np.stack(out)


def tracking_case(name, seed, nframes=6, steps=5):
    """Per-frame sequence (TrackWithIMU + TrackLocalMapWithIMU) of oracle/harness.py on one periodic synthetic stream."""
    from viorb_amd.synth import make_periodic_stream
    from oracle.harness import OracleTracker
    s = make_periodic_stream(seed, nframes)
    tr = OracleTracker(s["cam"], s["gw"], track_local_map=True)
    tr.bootstrap(s["frames"][0], s["pose_true"][0], s["t"][0], s["ns_true"][0], np.eye(12) * 1e3)
    rows, matches, locs = [], [], []
    for j in range(1, steps + 1):
        r = tr.step(s["frames"][j], s["imu"][j], s["t"][j], s["pose_true"][j])
        rows.append([r["nmatches"], r["n_inliers"], r["n_map"], r["n_loc"], r["n_obs2"], r["n_inliers2"], r["final_chi2"], r["final_chi2_2"]])
        matches.append(np.pad(r["match_after_discard"], (0, 1016 - len(r["match_after_discard"])), constant_values=-2))
        locs.append(np.pad(r["loc_match"], (0, 1016 - len(r["loc_match"])), constant_values=-2))
    np.savez_compressed(os.path.join(HERE, name), seed=seed, nframes=nframes, table=np.array(rows), ns2=np.stack([tr.last_ns]), match=np.stack(matches).astype(np.int32),
                        loc_match=np.stack(locs).astype(np.int32), marg=tr.marg_cov_inv)
    print(name, np.array(rows)[:, :6].astype(int).tolist())


def local_ba_cases():
    from viorb_amd.synth import make_local_ba_problem, make_local_ba_se3_problem
    p = make_local_ba_problem(1, W=10, n_points=600)
    r = ora.local_ba(p["kfs"], p["n_local"], p["prev_kf"], _preints(p), p["points"], p["edge_idx"], p["edge_obs"], p["gw"], p["cam"])
    np.savez_compressed(os.path.join(HERE, "local_ba_navstate_seed1.npz"), chi2=np.array([r["chi2_first"], r["chi2_final"]]), its=np.array([r["its_first"], r["its_second"]]),
                        erase=np.packbits(r["erase"]), n_edges=len(r["erase"]), kfs=r["kfs"], points_sum=r["points"].sum(0))
    q = make_local_ba_se3_problem(1, W=8, n_fixed=3, n_points=600)
    r2 = ora.local_ba_se3(q["kfs"], q["n_local"], q["points"], q["edge_idx"], q["edge_obs"], q["intr5"])
    np.savez_compressed(os.path.join(HERE, "local_ba_se3_seed1.npz"), chi2=np.array([r2["chi2_first"], r2["chi2_final"]]), its=np.array([r2["its_first"], r2["its_second"]]),
                        erase=np.packbits(r2["erase"]), n_edges=len(r2["erase"]), kfs=r2["kfs"], points_sum=r2["points"].sum(0))
    print("local_ba", r["its_first"], r["its_second"], r["chi2_final"], "| se3", r2["its_first"], r2["its_second"], r2["chi2_final"])


def matcher_cases():
    from viorb_amd.synth import make_vocabulary, descriptors_near_words, make_two_view_problem
    voc = make_vocabulary(11, 8, 5)
    desc = descriptors_near_words(12, voc, 800)
    t = ora.bow_transform(voc, desc, 4)
    p = make_two_view_problem(0, 900, 950, 500)
    n, m = ora.search_for_triangulation(p["k1"], p["d1"], p["hp1"], p["ur1"], p["node1"], p["k2"], p["d2"], p["hp2"], p["ur2"], p["node2"], p["F12"], p["Cw1"],
                                        p["pose2"], p["intr4"], p["sf"], p["level_sigma2"], False, True)
    nb, mb = ora.search_by_bow(p["d1"], p["k1"]["angle"], p["node1"], 1 - p["hp1"], p["d2"], p["k2"]["angle"], p["node2"], 0.7, True)
    np.savez_compressed(os.path.join(HERE, "matchers_seed0.npz"), bow_word=t["word"], bow_node=t["node"], bow_ids=t["bow_ids"], bow_vals=t["bow_vals"],
                        tri_n=n, tri_match=m.astype(np.int32), sbb_n=nb, sbb_match=mb.astype(np.int32))
    print("matchers", n, nb, len(t["bow_ids"]))


if __name__ == "__main__":
    tracking_case("track_seq_seed40.npz", 40)
    local_ba_cases()
    matcher_cases()
    extractor_case("extract_euroc_seed0.npz", 0, 752, 480, 1000)
    extractor_case("extract_euroc_seed1.npz", 1, 752, 480, 1000)
    extractor_case("extract_kitti_seed100.npz", 100, 1241, 376, 2000)
    extractor_case("extract_small_seed5.npz", 5, 160, 120, 300)

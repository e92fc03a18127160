"""Randomised differential run of the key-frame matchers, the BoW transform / search, the stereo association and the window solves against
the oracle: the comparisons of the fixed-seed tests in tests/ with random seeds and sizes (without those tests' workload-specific lower
bounds). `python tools/stress_frontend.py [cases per component]` on a GPU box; prints every failing case. A development aid."""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as oracle
oracle.lib()
import viorb_amd
from viorb_amd.synth import make_two_view_problem, make_vocabulary, descriptors_near_words, make_local_ba_problem, make_stereo_pair, KITTI_K
import test_gpu_bow as tb, test_gpu_local_ba as tl

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(31337)
fails, runs = 0, 0


def run(name, f, *a):
    global fails, runs
    runs += 1
    try:
        f(*a)
    except AssertionError as e:
        fails += 1; print("FAIL", name, a, str(e)[:300].replace("\n", " "))
    except Exception as e:
        fails += 1; print("ERROR", name, a, repr(e)[:300]); traceback.print_exc(limit=3)


def tri(seed, n1, n2, nc, stereo, only_stereo, ori):
    from viorb_amd import SearchForTriangulation
    p = make_two_view_problem(seed, n1, n2, nc, stereo_frac=stereo)
    args = (p["k1"], p["d1"], p["hp1"], p["ur1"], p["node1"], p["k2"], p["d2"], p["hp2"], p["ur2"], p["node2"], p["F12"], p["Cw1"], p["pose2"], p["intr4"],
            p["sf"], p["level_sigma2"], only_stereo, ori)
    n_ref, m_ref = oracle.search_for_triangulation(*args)
    n, m = SearchForTriangulation(*args)
    assert n == n_ref and np.array_equal(m, m_ref), (n, n_ref)


def bow_transform(seed, k, L, levelsup, n):
    from viorb_amd import ORBVocabulary
    voc = make_vocabulary(seed, k, L)
    desc = np.concatenate([descriptors_near_words(seed + 1, voc, n - n // 4), np.random.default_rng(seed + 2).integers(0, 256, (n // 4, 32), dtype=np.uint8)])
    ref = oracle.bow_transform(voc, desc, levelsup)
    V = ORBVocabulary(voc)
    word, weight, node = V.transform_features(desc, levelsup)
    ids, vals, fnode = V.transform(desc, levelsup)
    V.close()
    assert np.array_equal(word, ref["word"]) and np.array_equal(weight, ref["weight"]) and np.array_equal(node, ref["node"])
    assert np.array_equal(ids, ref["bow_ids"]) and np.array_equal(vals, ref["bow_vals"])


def bow_search(seed, k, L, nK, nF, ori):
    from viorb_amd import SearchByBoW
    voc = make_vocabulary(20 + seed, k, L)
    kd, ka, kn, kh, fd, fa, fn = tb._pair(oracle, voc, seed, nK, nF, max(1, min(nK, nF) * 2 // 3), 10)
    n_ref, m_ref = oracle.search_by_bow(kd, ka, kn, kh, fd, fa, fn, 0.7, ori)
    n, m = SearchByBoW(tb._kps(ka), kd, kn, kh, tb._kps(fa), fd, fn, 0.7, ori)
    assert n == n_ref and np.array_equal(m, m_ref), (n, n_ref)


def stereo(seed, w, h, nf):
    from viorb_amd.extractor import ComputeStereoMatches
    left, right, _ = make_stereo_pair(seed, w, h)
    gl, gr = viorb_amd.ORBextractor(nf, 1.2, 8, 20, 7), viorb_amd.ORBextractor(nf, 1.2, 8, 20, 7)
    kl, dl = gl(left); kr, dr = gr(right)
    ol, orr = oracle.Extractor(nf), oracle.Extractor(nf)
    okl, odl = ol(left); okr, odr = orr(right)
    assert np.array_equal(kl, okl) and np.array_equal(kr, okr)
    u, d, n = ComputeStereoMatches(gl, gr, KITTI_K["bf"], KITTI_K["fx"])
    ou, od, osad = oracle.stereo_match(ol, orr, okl, odl, okr, odr, KITTI_K["bf"], KITTI_K["fx"])
    assert np.array_equal(u[:len(ou)], ou) and np.array_equal(d[:len(od)], od) and n == int((ou >= 0).sum())


def lba(seed, W, npts, extra):
    from viorb_amd import LocalBundleAdjustmentNavState
    p = make_local_ba_problem(seed, W=W, n_points=npts, n_fixed_extra=extra)
    pre = tl._preints(oracle, p)
    ref = oracle.local_ba(*tl._args(p, pre))
    got = LocalBundleAdjustmentNavState(*tl._args(p, pre))
    assert (got["its_first"], got["its_second"]) == (ref["its_first"], ref["its_second"]), ((got["its_first"], got["its_second"]), (ref["its_first"], ref["its_second"]))
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-5 * ref["chi2_final"]
    assert np.array_equal(got["erase"], ref["erase"])
    np.testing.assert_allclose(got["kfs"], ref["kfs"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(got["points"], ref["points"], rtol=0, atol=1e-6)


def lba_se3(seed, W, nfix, npts, stereo):
    from viorb_amd import LocalBundleAdjustment
    from viorb_amd.synth import make_local_ba_se3_problem
    p = make_local_ba_se3_problem(seed, W=W, n_fixed=nfix, n_points=npts, stereo_frac=stereo)
    a = (p["kfs"], p["n_local"], p["points"], p["edge_idx"], p["edge_obs"], p["intr5"])
    ref = oracle.local_ba_se3(*a); got = LocalBundleAdjustment(*a)
    assert (got["its_first"], got["its_second"]) == (ref["its_first"], ref["its_second"])
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-5 * ref["chi2_final"] and np.array_equal(got["erase"], ref["erase"])
    np.testing.assert_allclose(got["kfs"], ref["kfs"], rtol=0, atol=1e-7)


def undistort(seed):
    r = np.random.default_rng(seed)
    K = np.float32([r.uniform(200, 900), r.uniform(200, 900), r.uniform(300, 700), r.uniform(200, 400)])
    D = np.float32([r.uniform(-0.4, 0.3), r.uniform(-0.2, 0.2), r.uniform(-3e-3, 3e-3), r.uniform(-3e-3, 3e-3), r.choice([0.0, r.uniform(-0.05, 0.05)])])
    if D[0] == 0: D[0] = np.float32(0.01)
    xy = r.uniform([-200, -200], [1500, 1000], (int(r.integers(1, 30000)), 2)).astype(np.float32)
    assert np.array_equal(viorb_amd.UndistortKeyPoints(xy, K, D).view(np.uint32), oracle.undistort_points(xy, K, D).view(np.uint32))
    w, h = int(r.integers(100, 2000)), int(r.integers(100, 1200))
    assert np.array_equal(viorb_amd.ComputeImageBounds(w, h, K, D), oracle.image_bounds(w, h, K, D))


for i in range(N):
    s = int(rng.integers(100, 100000))
    run("local_ba_se3", lba_se3, s, int(rng.integers(1, 41)), int(rng.integers(1, 4)), int(rng.integers(30, 1200)), float(rng.choice([0.0, 0.5, 1.0])))
    run("undistort", undistort, s)
    n1, n2 = int(rng.integers(2, 2500)), int(rng.integers(2, 2500)); nc = int(rng.integers(1, min(n1, n2) + 1))
    run("triangulation", tri, s, n1, n2, nc, float(rng.choice([0.0, 0.3, 1.0])), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    k, L = int(rng.integers(2, 18)), int(rng.integers(2, 7))
    while k ** L > 1_200_000: L -= 1
    run("bow_transform", bow_transform, s, k, L, int(rng.integers(1, 7)), int(rng.integers(4, 2500)))
    k, L = int(rng.integers(3, 11)), int(rng.integers(3, 7))
    while k ** L > 1_200_000: L -= 1
    run("bow_search", bow_search, s % 1000, k, L, int(rng.integers(2, 2200)), int(rng.integers(2, 2200)), bool(rng.integers(0, 2)))
    if i % 3 == 0:
        run("stereo", stereo, s, int(rng.integers(300, 1300)), int(rng.integers(200, 500)), int(rng.integers(300, 2200)))
    run("local_ba", lba, s, int(rng.integers(1, 21)), int(rng.integers(40, 1500)), int(rng.integers(2, 5)))
print("runs", runs, "failures", fails)

"""GPU parity (bit-exact) of ORBmatcher::SearchForTriangulation and ORBmatcher::Fuse (csrc/bow_matcher.hip, frontend_kernels.hip)
against the oracle (oracle/kf_matcher.cpp)."""
import numpy as np
import pytest
from viorb_amd.synth import make_two_view_problem, local_points_f32

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n1,n2,nc,stereo,only_stereo,ori", [(0, 900, 950, 500, 0.0, False, True), (1, 1000, 1000, 600, 0.4, False, False),
                                                                    (2, 2000, 1900, 1100, 0.6, True, True), (3, 40, 1200, 30, 0.0, False, True),
                                                                    (4, 700, 1, 1, 0.0, False, True)])
def test_search_for_triangulation_bit_exact(oracle, seed, n1, n2, nc, stereo, only_stereo, ori):
    from viorb_amd import SearchForTriangulation
    p = make_two_view_problem(seed, n1, n2, nc, stereo_frac=stereo)
    args = (p["k1"], p["d1"], p["hp1"], p["ur1"], p["node1"], p["k2"], p["d2"], p["hp2"], p["ur2"], p["node2"], p["F12"], p["Cw1"], p["pose2"], p["intr4"],
            p["sf"], p["level_sigma2"], only_stereo, ori)
    n_ref, m_ref = oracle.search_for_triangulation(*args)
    n, m = SearchForTriangulation(*args)
    assert n == n_ref and np.array_equal(m, m_ref)
    if nc >= 500:
        assert n > 50


def _fuse_problem(seed, stereo):
    p = make_two_view_problem(seed, 1000, 1100, 700, stereo_frac=stereo)
    rng = np.random.default_rng(seed)
    # map points: the common 3-D points (created from key frame 1) + random ones; descriptors = key frame 2's nearest keypoint's, with noise
    R2, t2 = p["pose2"][:9].reshape(3, 3).astype(np.float64), p["pose2"][9:].astype(np.float64)
    fx, fy, cx, cy = [float(v) for v in p["intr4"]]
    X = np.concatenate([p["X"], np.stack([rng.uniform(-4, 4, 300), rng.uniform(-3, 3, 300), rng.uniform(1, 15, 300)], 1)])
    uv2 = (R2 @ X.T).T + t2; uv2 = np.stack([fx * uv2[:, 0] / uv2[:, 2] + cx, fy * uv2[:, 1] / uv2[:, 2] + cy], 1)
    k2xy = np.stack([p["k2"]["x"], p["k2"]["y"]], 1).astype(np.float64)
    partner = np.array([int(np.argmin(((k2xy - q) ** 2).sum(1))) for q in uv2])
    pts_f = local_points_f32(p["k2"]["octave"][partner], p["pose1"].astype(np.float64), X.astype(np.float32), p["sf"])
    valid = (rng.random(len(X)) < 0.9).astype(np.uint8)
    desc = p["d2"][partner].copy()
    for _ in range(8):
        b = rng.integers(0, 256, len(X)); desc[np.arange(len(X)), b >> 3] ^= (1 << (b & 7)).astype(np.uint8)
    intr5 = np.concatenate([p["intr4"], [np.float32(40.0)]]).astype(np.float32)
    return p, pts_f, valid, desc, intr5


@pytest.mark.parametrize("seed,stereo,th", [(5, 0.0, 3.0), (6, 0.5, 3.0), (7, 0.0, 10.0)])
def test_fuse_bit_exact(oracle, seed, stereo, th):
    from viorb_amd import Fuse
    p, pts_f, valid, desc, intr5 = _fuse_problem(seed, stereo)
    log_sf = np.float32(np.log(np.float64(p["sf"][1])))
    bounds = (0.0, 752.0, 0.0, 480.0)
    n_ref, b_ref = oracle.fuse(p["k2"], p["d2"], p["ur2"], bounds, p["pose2"], intr5, p["sf"], p["inv_level_sigma2"], log_sf, pts_f, valid, desc, th)
    n, b = Fuse(p["k2"], p["d2"], p["ur2"], bounds, p["pose2"], intr5, p["sf"], p["inv_level_sigma2"], pts_f, valid, desc, th)
    assert n == n_ref and np.array_equal(b, b_ref)
    assert n > 100

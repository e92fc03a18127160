"""GPU parity of Optimizer::LocalBundleAdjustmentNavState (viorb_local_ba_navstate, csrc/local_ba.hip) against the oracle
(oracle/local_ba.cpp): same LM trajectory (iteration counts), chi2 within 1e-5 relative, identical erase flags, key-frame
states and points to solver precision; stop flag honoured; argument checks."""
import numpy as np
import pytest
from viorb_amd.synth import make_local_ba_problem

pytestmark = pytest.mark.gpu


def _preints(oracle, p):
    out = []
    for i, (imu, t0, t1) in enumerate(p["imu"]):
        j = i - 1 if i > 0 else p["prev_kf"]
        out.append(oracle.preintegrate(imu, p["kfs"][j][10:13], p["kfs"][j][13:16], t0, t1))
    return np.stack(out)


def _args(p, pre):
    return (p["kfs"], p["n_local"], p["prev_kf"], pre, p["points"], p["edge_idx"], p["edge_obs"], p["gw"], p["cam"])


@pytest.mark.parametrize("seed,W,npts,extra", [(1, 10, 600, 3), (2, 4, 80, 2), (3, 20, 2000, 3), (4, 1, 60, 2)])
def test_local_ba_matches_oracle(oracle, seed, W, npts, extra):
    from viorb_amd import LocalBundleAdjustmentNavState
    p = make_local_ba_problem(seed, W=W, n_points=npts, n_fixed_extra=extra)
    pre = _preints(oracle, p)
    ref = oracle.local_ba(*_args(p, pre))
    got = LocalBundleAdjustmentNavState(*_args(p, pre))
    assert (got["its_first"], got["its_second"]) == (ref["its_first"], ref["its_second"])
    assert abs(got["chi2_first"] - ref["chi2_first"]) <= 1e-5 * ref["chi2_first"]        # north_star tolerance
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-5 * ref["chi2_final"]
    assert np.array_equal(got["erase"], ref["erase"])
    np.testing.assert_allclose(got["kfs"], ref["kfs"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(got["points"], ref["points"], rtol=0, atol=1e-6)


def test_local_ba_batch_equals_single_calls(oracle):
    """viorb_local_ba_navstate_batch: windows of different sizes kept in flight together give what the single
    calls give (same kernels; the Schur accumulation's LDS atomics make two runs differ in the last bits), in input order."""
    from viorb_amd import LocalBundleAdjustmentNavState, LocalBundleAdjustmentNavStateBatch
    probs = []
    for seed, W, npts in [(11, 6, 300), (12, 10, 500), (13, 3, 90), (14, 8, 400), (15, 6, 300), (16, 12, 700), (17, 2, 60), (18, 5, 200), (19, 9, 450)]:
        p = make_local_ba_problem(seed, W=W, n_points=npts, n_fixed_extra=2)
        a = _args(p, _preints(oracle, p))
        probs.append(dict(kfs=a[0], n_local=a[1], prev_kf=a[2], preint=a[3], points=a[4], edge_idx=a[5], edge_obs=a[6], gw=a[7], cam=a[8]))
    single = [LocalBundleAdjustmentNavState(**q) for q in probs]
    for in_flight in (1, 4, 16):
        batch = LocalBundleAdjustmentNavStateBatch(probs, max_in_flight=in_flight)
        assert len(batch) == len(single)
        for g, r in zip(batch, single):
            assert (g["its_first"], g["its_second"]) == (r["its_first"], r["its_second"])
            np.testing.assert_array_equal(g["erase"], r["erase"])
            np.testing.assert_allclose(g["kfs"], r["kfs"], rtol=0, atol=1e-9)          # LDS atomics: run-to-run rounding differences
            np.testing.assert_allclose(g["points"], r["points"], rtol=0, atol=1e-9)
            assert abs(g["chi2_final"] - r["chi2_final"]) <= 1e-9 * r["chi2_final"]


def test_local_ba_batch_stop_flag_set_before_the_call(oracle):
    """A window whose pbStopFlag is already set comes back untouched from the lock-step batch while its neighbours are solved."""
    import ctypes as C
    from viorb_amd import capi, LocalBundleAdjustmentNavState
    from viorb_amd.capi import lib, check
    probs = []
    for seed in (21, 22, 23):
        p = make_local_ba_problem(seed, W=5, n_points=200, n_fixed_extra=2)
        probs.append(_args(p, _preints(oracle, p)))
    single = [LocalBundleAdjustmentNavState(*a) for a in probs]
    W = (capi.LbaWindow * 3)()
    keep, stop = [], np.ones(1, np.int32)
    for i, a in enumerate(probs):
        kfs = np.ascontiguousarray(a[0], np.float64); pre = np.ascontiguousarray(a[3], np.float64); pts = np.ascontiguousarray(a[4], np.float64)
        ei = np.ascontiguousarray(a[5], np.int32); eo = np.ascontiguousarray(a[6], np.float64); gw = np.ascontiguousarray(a[7], np.float64); cam = np.ascontiguousarray(a[8], np.float64)
        ko, po, er, info = np.zeros((a[1], 22)), np.zeros_like(pts), np.zeros(len(ei), np.uint8), np.zeros(6)
        keep.append((kfs, pre, pts, ei, eo, gw, cam, ko, po, er, info))
        w = W[i]
        w.kfs = kfs.ctypes.data; w.nk = len(kfs); w.n_local = a[1]; w.prev_kf = a[2]; w.preint = pre.ctypes.data; w.points = pts.ctypes.data; w.np = len(pts)
        w.edge_idx = ei.ctypes.data; w.edge_obs = eo.ctypes.data; w.ne = len(ei); w.gw = gw.ctypes.data; w.cam = cam.ctypes.data
        w.stop = stop.ctypes.data if i == 1 else None
        w.kfs_out = ko.ctypes.data; w.points_out = po.ctypes.data; w.erase = er.ctypes.data; w.info = info.ctypes.data
    check(lib().viorb_local_ba_navstate_batch(C.cast(W, C.c_void_p), 3, 0))
    assert [W[i].status for i in range(3)] == [0, 0, 0]
    np.testing.assert_array_equal(keep[1][7], keep[1][0][:probs[1][1]])            # stopped window: key frames unchanged
    np.testing.assert_array_equal(keep[1][8], keep[1][2]); assert keep[1][9].sum() == 0 and keep[1][10][2] == 0
    for i in (0, 2):
        assert (int(keep[i][10][2]), int(keep[i][10][3])) == (single[i]["its_first"], single[i]["its_second"])
        np.testing.assert_allclose(keep[i][7], single[i]["kfs"], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(keep[i][9], single[i]["erase"])


def test_local_ba_without_prev_keyframe(oracle):
    """prev_kf = -1: the first local key frame has no IMU / bias factor (map start)."""
    from viorb_amd import LocalBundleAdjustmentNavState
    p = make_local_ba_problem(5, W=6, n_points=300, n_fixed_extra=2)
    pre = _preints(oracle, p)
    a = list(_args(p, pre)); a[2] = -1
    ref = oracle.local_ba(*a)
    got = LocalBundleAdjustmentNavState(*a)
    assert (got["its_first"], got["its_second"]) == (ref["its_first"], ref["its_second"])
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-5 * ref["chi2_final"]
    assert np.array_equal(got["erase"], ref["erase"])
    np.testing.assert_allclose(got["kfs"], ref["kfs"], rtol=0, atol=1e-7)


def test_local_ba_stop_flag_raised_during_the_solve(oracle):
    """pbStopFlag goes up from another thread while the window is being solved (LocalMapping::InterruptBA): the call returns the state after
    the trial in flight — fewer Levenberg iterations than the full 5 + 10, a chi2 no worse than the start, finite outputs."""
    import threading, time
    from viorb_amd import LocalBundleAdjustmentNavState
    p = make_local_ba_problem(3, W=20, n_points=2000)
    pre = _preints(oracle, p)
    full = LocalBundleAdjustmentNavState(*_args(p, pre))
    for delay in (0.0, 0.0005, 0.0015):
        stop = np.zeros(1, np.int32)
        t = threading.Thread(target=lambda: (time.sleep(delay), stop.__setitem__(0, 1)))
        t.start()
        got = LocalBundleAdjustmentNavState(*_args(p, pre), stop=stop)
        t.join()
        assert np.isfinite(got["kfs"]).all() and np.isfinite(got["points"]).all()
        assert got["its_first"] + got["its_second"] <= full["its_first"] + full["its_second"]
        if got["its_first"] + got["its_second"] == 0:
            assert np.array_equal(got["points"], p["points"])


def test_local_ba_stop_flag_and_argument_checks(oracle):
    from viorb_amd import LocalBundleAdjustmentNavState, ViorbError
    p = make_local_ba_problem(6, W=5, n_points=200)
    pre = _preints(oracle, p)
    stop = np.ones(1, np.int32)
    got = LocalBundleAdjustmentNavState(*_args(p, pre), stop=stop)           # pbStopFlag already set: nothing moves (:2023-2026)
    assert np.array_equal(got["kfs"], p["kfs"][:p["n_local"]]) and np.array_equal(got["points"], p["points"])
    assert got["erase"].sum() == 0 and got["its_first"] == 0
    bad = p["edge_idx"].copy(); bad[0, 1] = 99
    with pytest.raises(ViorbError):
        LocalBundleAdjustmentNavState(p["kfs"], p["n_local"], p["prev_kf"], pre, p["points"], bad, p["edge_obs"], p["gw"], p["cam"])
    with pytest.raises(ViorbError):                                           # edges must be sorted by point
        LocalBundleAdjustmentNavState(p["kfs"], p["n_local"], p["prev_kf"], pre, p["points"], p["edge_idx"][::-1], p["edge_obs"][::-1], p["gw"], p["cam"])


@pytest.mark.parametrize("seed,W,nfix,npts,stereo", [(1, 8, 3, 600, 0.5), (2, 4, 2, 120, 0.0), (3, 20, 5, 2000, 0.7), (4, 40, 4, 1500, 0.3), (5, 1, 3, 80, 1.0)])
def test_local_ba_se3_matches_oracle(oracle, seed, W, nfix, npts, stereo):
    """Vision-only LocalBundleAdjustment (viorb_local_ba_se3) against oracle/local_ba_se3.cpp."""
    from viorb_amd import LocalBundleAdjustment
    from viorb_amd.synth import make_local_ba_se3_problem
    p = make_local_ba_se3_problem(seed, W=W, n_fixed=nfix, n_points=npts, stereo_frac=stereo)
    a = (p["kfs"], p["n_local"], p["points"], p["edge_idx"], p["edge_obs"], p["intr5"])
    ref = oracle.local_ba_se3(*a)
    got = LocalBundleAdjustment(*a)
    assert (got["its_first"], got["its_second"]) == (ref["its_first"], ref["its_second"])
    assert abs(got["chi2_first"] - ref["chi2_first"]) <= 1e-5 * ref["chi2_first"]
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-5 * ref["chi2_final"]
    assert np.array_equal(got["erase"], ref["erase"])
    np.testing.assert_allclose(got["kfs"], ref["kfs"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(got["points"], ref["points"], rtol=0, atol=1e-5)
    stop = np.ones(1, np.int32)
    g2 = LocalBundleAdjustment(*a, stop=stop)
    assert np.array_equal(g2["kfs"], p["kfs"][:W]) and g2["its_first"] == 0


def test_local_ba_se3_batch_equals_single_calls():
    """viorb_local_ba_se3_batch: several vision-only windows in flight give what the single calls give (to the rounding noise of the
    Schur accumulation's LDS atomics), in input order."""
    from viorb_amd import LocalBundleAdjustment, LocalBundleAdjustmentBatch
    from viorb_amd.synth import make_local_ba_se3_problem
    probs = []
    for seed, W, nfix, npts, stereo in [(21, 6, 2, 300, 0.0), (22, 10, 3, 500, 0.5), (23, 3, 1, 90, 1.0), (24, 8, 2, 400, 0.3), (25, 12, 3, 600, 0.0)]:
        p = make_local_ba_se3_problem(seed, W=W, n_fixed=nfix, n_points=npts, stereo_frac=stereo)
        probs.append(dict(kfs=p["kfs"], n_local=p["n_local"], points=p["points"], edge_idx=p["edge_idx"], edge_obs=p["edge_obs"], intr5=p["intr5"]))
    single = [LocalBundleAdjustment(**q) for q in probs]
    for in_flight in (1, 3, 16):
        batch = LocalBundleAdjustmentBatch(probs, max_in_flight=in_flight)
        for g, r in zip(batch, single):
            assert (g["its_first"], g["its_second"]) == (r["its_first"], r["its_second"])
            np.testing.assert_array_equal(g["erase"], r["erase"])
            np.testing.assert_allclose(g["kfs"], r["kfs"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(g["points"], r["points"], rtol=0, atol=1e-8)
            assert abs(g["chi2_final"] - r["chi2_final"]) <= 1e-9 * r["chi2_final"]

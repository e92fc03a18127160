"""-m gpu, needs TWO devices (skipped on the one-GPU test box): handles created on device 1 of the same process give what device 0 gives —
the per-kernel dynamic-LDS limit is raised per (device, kernel) (viorb_amd/csrc/orb_extractor.hip: raise_dynamic_lds), every handle takes its
device, and the window solve follows viorb_local_ba_set_device. What `bench.py` relies on when the driver gives every rank its own GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _need_two():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")


def test_extractor_and_tracker_on_the_second_device():
    _need_two()
    import torch
    import viorb_amd
    from viorb_amd.synth import make_image, make_periodic_stream
    from viorb_amd.tracker import NativeTracker
    img = make_image(3, 752, 480)
    k0, d0 = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7, device=0)(img)
    k1, d1 = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7, device=1)(img)
    np.testing.assert_array_equal(k0, k1); np.testing.assert_array_equal(d0, d1)
    s = make_periodic_stream(21, 3)
    res = []
    for dev_index in (0, 1):
        dev = torch.device("cuda", dev_index)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        tr = NativeTracker(s["cam"], s["gw"], 1, 752, 480, 1000, track_local_map=True, device=dev_index)
        tr.bootstrap(up(s["frames"][0][None]), up(s["pose_true"][0][None]), up(np.array([s["t"][0]])), up(s["ns_true"][0][None]), up((np.eye(12) * 1e3).ravel()[None]))
        for j in (1, 2):
            tr.step(up(s["frames"][j][None]), up(s["imu"][j][None]), up(np.array([s["t"][j]])), up(s["pose_true"][j][None]))
        g = tr.results(["state", "status", "nmatches", "final_ns", "info2"])
        res.append(g)
    for key in ("state", "status", "nmatches", "info2"):
        np.testing.assert_array_equal(res[0][key], res[1][key])
    np.testing.assert_allclose(res[0]["final_ns"], res[1]["final_ns"], rtol=0, atol=1e-12)
    assert int(res[0]["state"][0]) == 0 and int(res[0]["status"][0]) == 0


def test_window_solve_on_the_second_device():
    _need_two()
    import viorb_amd
    from viorb_amd import LocalBundleAdjustment
    from viorb_amd.synth import make_local_ba_se3_problem
    p = make_local_ba_se3_problem(5, W=6, n_fixed=2, n_points=300, stereo_frac=0.3)
    a = (p["kfs"], p["n_local"], p["points"], p["edge_idx"], p["edge_obs"], p["intr5"])
    L = viorb_amd.lib()
    try:
        assert L.viorb_local_ba_set_device(0) == 0
        r0 = LocalBundleAdjustment(*a)
        assert L.viorb_local_ba_set_device(1) == 0
        r1 = LocalBundleAdjustment(*a)
    finally:
        L.viorb_local_ba_set_device(-1)
    assert (r0["its_first"], r0["its_second"]) == (r1["its_first"], r1["its_second"]) and np.array_equal(r0["erase"], r1["erase"])
    np.testing.assert_allclose(r0["kfs"], r1["kfs"], rtol=0, atol=1e-9)

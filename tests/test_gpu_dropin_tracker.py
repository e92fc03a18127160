"""-m gpu: ONE stream through the host-buffer drop-ins, call by call (viorb_amd.tracker.DropinTracker: viorb_extract, viorb_undistort_points,
viorb_preintegrate, viorb_search_by_projection_frame, viorb_pose_opt_vi, viorb_search_by_projection_points, viorb_pose_opt_vi — what a VIORB
Tracking thread calls through viorb_amd/shim/) against the oracle twin on the same synthetic stream, frame by frame, with the pinhole camera
and with the EuRoC lens (Examples/ROS/ORB_VIO/launch/euroc.yaml:64-67). It is the sequence `bench.py --config dropin` times."""
import numpy as np
import pytest
import viorb_amd
from viorb_amd.synth import make_periodic_stream, EUROC_DIST

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dist", [None, EUROC_DIST], ids=["pinhole", "euroc_lens"])
def test_dropin_sequence_equals_oracle_twin(dist):
    if viorb_amd.lib().viorb_device_count() < 1:
        pytest.fail("no HIP device visible")
    from viorb_amd.tracker import DropinTracker
    from oracle.harness import OracleTracker
    F = 6
    s = make_periodic_stream(77, F, dist=dist)
    mci = np.eye(12) * 1e3
    tr = DropinTracker(s["cam"], s["gw"], dist_coef=dist)
    tw = OracleTracker(s["cam"], s["gw"], track_local_map=True, dist_coef=dist)
    tr.bootstrap(s["frames"][0], s["pose_true"][0], s["t"][0], s["ns_true"][0], mci)
    tw.bootstrap(s["frames"][0], s["pose_true"][0], s["t"][0], s["ns_true"][0], mci)
    for k in range(1, 10):                                   # across the key-frame boundary (j == 0) and the mbMapUpdated frame after it
        j = k % F
        if j == 0:
            kw = dict(t_next_last=0.0, reset_ns=s["ns_true"][0], reset_marg=mci)
            a = tr.step(s["frames"][0], s["imu"][0], s["period"], s["pose_true"][0], **kw)
            b = tw.step(s["frames"][0], s["imu"][0], s["period"], s["pose_true"][0], **kw)
        else:
            mu = j == 1 and k > 1
            a = tr.step(s["frames"][j], s["imu"][j], s["t"][j], s["pose_true"][j], map_updated=mu)
            b = tw.step(s["frames"][j], s["imu"][j], s["t"][j], s["pose_true"][j], map_updated=mu)
        assert a["state"] == b["state"] == 0, (k, a["state"], b["state"])
        assert a["nmatches"] == b["nmatches"] and a["inliers"] == b["inliers"], (k, a["nmatches"], b["nmatches"], a["inliers"], b["inliers"])
        np.testing.assert_allclose(a["final_ns"], b["final_ns"], rtol=0, atol=1e-7, err_msg="frame %d" % k)
        np.testing.assert_allclose(tr.marg_cov_inv, tw.marg_cov_inv, rtol=1e-6, atol=1e-6 * np.abs(tw.marg_cov_inv).max())
        assert np.linalg.norm(b["final_ns"][:3] - s["ns_true"][j][:3]) < 0.05
    if dist is not None:
        assert "viorb_undistort_points" in tr.times and tr.times["viorb_undistort_points"][1] == 10      # bootstrap + 9 frames

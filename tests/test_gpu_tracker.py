"""-m gpu end-to-end parity: the batched device tracking sequence (viorb_amd.tracker.BatchedTracker: extract ->
grid -> IMU predict -> SearchByProjection -> PoseOptimization, chained over frames) against the oracle-side twin
(oracle/harness.py) on the same synthetic periodic streams, step by step."""
import numpy as np
import pytest
import viorb_amd
from viorb_amd.synth import make_periodic_stream

pytestmark = pytest.mark.gpu


def test_tracking_sequence_matches_oracle_step_by_step():
    if viorb_amd.lib().viorb_device_count() < 1:
        pytest.fail("no HIP device visible")
    import torch
    from viorb_amd.tracker import BatchedTracker
    from oracle.harness import OracleTracker
    F, B = 6, 2
    streams = [make_periodic_stream(40 + b, F) for b in range(B)]
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    frames = up(np.stack([s["frames"] for s in streams], 1)); imu = up(np.stack([s["imu"] for s in streams], 1))
    t_frames = up(np.stack([s["t"] for s in streams], 1)); t_period = up(np.array([s["period"] for s in streams]))
    pose_true = up(np.stack([s["pose_true"] for s in streams], 1)); ns_true = up(np.stack([s["ns_true"] for s in streams], 1))
    mci0 = up(np.stack([np.eye(12).ravel() * 1e3] * B))
    tr = BatchedTracker(streams[0]["cam"], streams[0]["gw"], B)
    tr.bootstrap(frames[0], pose_true[0], t_frames[0], ns_true[0], mci0)
    ors = []
    for s in streams:
        o = OracleTracker(s["cam"], s["gw"])
        o.bootstrap(s["frames"][0], s["pose_true"][0], s["t"][0], s["ns_true"][0], np.eye(12) * 1e3)
        ors.append(o)
    zeros = torch.zeros(B, dtype=torch.float64, device=dev)
    for k in range(1, 9):                                   # includes the loop-closing step (j == 0) and one more lap
        j = k % F
        if j == 0:
            tr.step(frames[0], imu[0], t_period, pose_true[0], t_next_last=zeros)
        else:
            tr.step(frames[j], imu[j], t_frames[j], pose_true[j])
        torch.cuda.synchronize()
        info, nm, match = tr.info.cpu().numpy(), tr.nmatches.cpu().numpy(), tr.cur_match.cpu().numpy()
        out_ns, cur_ns = tr.out_ns.cpu().numpy(), tr.cur_ns.cpu().numpy()
        for b, s in enumerate(streams):
            r = ors[b].step(s["frames"][j], s["imu"][j], s["t"][j] if j else s["period"], s["pose_true"][j], t_next_last=0.0 if j == 0 else None)
            assert nm[b] == r["nmatches"], (k, b)
            np.testing.assert_array_equal(match[b, :r["n_kps"]], r["match"], err_msg="step %d stream %d" % (k, b))
            np.testing.assert_allclose(cur_ns[b], r["pred_ns"], rtol=0, atol=1e-9)
            assert int(info[b, 0]) == r["n_inliers"], (k, b)
            assert abs(info[b, 1] - r["final_chi2"]) <= 1e-5 * r["final_chi2"], (k, b, info[b, 1], r["final_chi2"])
            np.testing.assert_allclose(out_ns[b], r["ns"], rtol=0, atol=1e-7)
            assert r["n_inliers"] > 300 and np.linalg.norm(r["ns"][:3] - s["ns_true"][j][:3]) < 0.05


def test_track_local_map_stage_matches_oracle_step_by_step():
    """TrackWithIMU + TrackLocalMapWithIMU per frame (discard outliers -> SearchLocalPoints -> second PoseOptimization with the
    marginal), chained over frames and across a key-frame boundary, against the oracle twin."""
    import torch
    from viorb_amd.tracker import BatchedTracker
    from oracle.harness import OracleTracker
    F, B = 6, 2
    streams = [make_periodic_stream(60 + b, F) for b in range(B)]
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    frames = up(np.stack([s["frames"] for s in streams], 1)); imu = up(np.stack([s["imu"] for s in streams], 1))
    t_frames = up(np.stack([s["t"] for s in streams], 1)); t_period = up(np.array([s["period"] for s in streams]))
    pose_true = up(np.stack([s["pose_true"] for s in streams], 1)); ns_true = up(np.stack([s["ns_true"] for s in streams], 1))
    mci0 = up(np.stack([np.eye(12).ravel() * 1e3] * B))
    tr = BatchedTracker(streams[0]["cam"], streams[0]["gw"], B, track_local_map=True)
    tr.bootstrap(frames[0], pose_true[0], t_frames[0], ns_true[0], mci0)
    ors = []
    for s in streams:
        o = OracleTracker(s["cam"], s["gw"], track_local_map=True)
        o.bootstrap(s["frames"][0], s["pose_true"][0], s["t"][0], s["ns_true"][0], np.eye(12) * 1e3)
        ors.append(o)
    zeros = torch.zeros(B, dtype=torch.float64, device=dev)
    cap = tr.cap
    total_local = 0
    for k in range(1, 10):
        j = k % F
        if j == 0:          # the loop closes: the harness's key-frame boundary (bench.py does the same every 8 frames)
            tr.step(frames[0], imu[0], t_period, pose_true[0], t_next_last=zeros, chain_estimate=False, true_ns=ns_true[0], marg_reset=mci0)
        else:
            tr.step(frames[j], imu[j], t_frames[j], pose_true[j])
        torch.cuda.synchronize()
        g = lambda t: t.cpu().numpy()
        info, nm, match, out_ns = g(tr.info), g(tr.nmatches), g(tr.cur_match), g(tr.out_ns)
        n_map, n_loc, loc_match, info2, out_ns2, n_obs2 = g(tr.n_map), g(tr.n_loc), g(tr.loc_match), g(tr.info2), g(tr.out_ns2), g(tr.n_cur2)
        assert (g(tr.status) == 0).all() and (g(tr.status2) == 0).all()
        for b, s in enumerate(streams):
            if j:
                r = ors[b].step(s["frames"][j], s["imu"][j], s["t"][j], s["pose_true"][j])
            else:
                r = ors[b].step(s["frames"][0], s["imu"][0], s["period"], s["pose_true"][0], t_next_last=0.0, reset_ns=s["ns_true"][0],
                                reset_marg=np.eye(12) * 1e3)
            n = r["n_kps"]
            assert nm[b] == r["nmatches"] and int(info[b, 0]) == r["n_inliers"], (k, b)
            np.testing.assert_array_equal(match[b, :n], r["match_after_discard"], err_msg="discard, step %d stream %d" % (k, b))
            np.testing.assert_allclose(out_ns[b], r["ns"], rtol=0, atol=1e-7)
            assert n_map[b] == r["n_map"] and n_loc[b] == r["n_loc"], (k, b, n_map[b], r["n_map"], n_loc[b], r["n_loc"])
            # device local index = slot * cap + i, oracle = offsets[slot] + i
            lm = loc_match[b, :n]
            offs = r["loc_offsets"]
            conv = np.where(lm >= 0, offs[np.minimum(np.maximum(lm, 0) // cap, len(offs) - 1)] + np.maximum(lm, 0) % cap, -1)
            np.testing.assert_array_equal(conv, r["loc_match"], err_msg="local match, step %d stream %d" % (k, b))
            assert n_obs2[b] == r["n_obs2"] and int(info2[b, 0]) == r["n_inliers2"], (k, b)
            assert abs(info2[b, 1] - r["final_chi2_2"]) <= 1e-5 * r["final_chi2_2"], (k, b)
            np.testing.assert_allclose(out_ns2[b], r["ns2"], rtol=0, atol=1e-7)
            total_local += r["n_loc"]
            assert r["n_inliers2"] > 300 and np.linalg.norm(r["ns2"][:3] - s["ns_true"][j][:3]) < 0.05
    assert total_local > 50                                     # the local-map search contributes matches once the map has history

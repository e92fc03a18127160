"""-m gpu parity tests of the tracking front-end kernels (grid, IMU prediction, SearchByProjection, pose
optimisation) against the CPU oracle, through the C ABI. Integer / index results bit-exact; FP64 solver
results within the north-star tolerance (final cost 1e-5 relative)."""
import numpy as np
import pytest
import viorb_amd
from viorb_amd.synth import make_vi_stream, make_vio_problem, backproject_to_plane, cam_pose_from_navstate

pytestmark = pytest.mark.gpu
BOUNDS = (0.0, 752.0, 0.0, 480.0)


@pytest.fixture(scope="module")
def torch_cuda():
    if viorb_amd.lib().viorb_device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X (and never fall back)")
    import torch
    return torch


@pytest.fixture(scope="module")
def streams(oracle):
    """Three 2-frame streams with oracle features for both frames and map points for frame 0."""
    out = []
    ex = oracle.Extractor()
    for seed in (1, 2, 3):
        s = make_vi_stream(seed, 2)
        k0, d0 = ex(s["frames"][0]); k1, d1 = ex(s["frames"][1])
        Pw = backproject_to_plane(np.stack([k0["x"], k0["y"]], 1).astype(np.float64), s["ns_true"][0], s["cam"]).astype(np.float32)
        rng = np.random.default_rng(seed)
        flags = np.full(len(k0), 1 | 4, np.uint8)
        flags[rng.random(len(k0)) < 0.2] = 0
        flags[rng.random(len(k0)) < 0.05] |= 2
        flags[rng.random(len(k0)) < 0.1] &= ~np.uint8(4)
        out.append(dict(s=s, k0=k0, d0=d0, k1=k1, d1=d1, Pw=Pw, flags=flags, tables=ex.tables()))
    return out


def make_frontend(st, B, cap):
    t = st["tables"]
    return viorb_amd.Frontend(st["s"]["cam"], st["s"]["gw"], t["scale"], t["inv_sigma2"], BOUNDS, max_batch=B, cap=cap)


def pad(torch, arrs, cap, dtype, tail=()):
    out = np.zeros((len(arrs), cap) + tuple(tail), dtype)
    for b, a in enumerate(arrs):
        out[b, :len(a)] = a
    return torch.from_numpy(out.view(np.uint8).reshape(out.shape + (-1,)) if dtype == viorb_amd.KP_DTYPE else out).cuda()


def test_grid_matches_oracle(torch_cuda, oracle, streams):
    torch = torch_cuda
    B, cap = len(streams), 1016
    fe = make_frontend(streams[0], B, cap)
    kps = pad(torch, [s["k1"] for s in streams], cap, viorb_amd.KP_DTYPE)
    cnt = torch.tensor([len(s["k1"]) for s in streams], dtype=torch.int32, device="cuda")
    cs = torch.zeros((B, 3073), dtype=torch.int32, device="cuda"); ci = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    fe.grid(kps.data_ptr(), cnt.data_ptr(), B, cs, ci)
    torch.cuda.synchronize()
    for b, s in enumerate(streams):
        ocs, oci = oracle.frame_grid(s["k1"], BOUNDS)
        np.testing.assert_array_equal(cs[b].cpu().numpy(), ocs)
        np.testing.assert_array_equal(ci[b, :len(oci)].cpu().numpy(), oci)


def test_imu_predict_matches_oracle(torch_cuda, oracle, streams):
    torch = torch_cuda
    B = len(streams)
    fe = make_frontend(streams[0], B, 64)
    imu = torch.from_numpy(np.stack([s["s"]["imu"][1] for s in streams])).cuda()
    tl = torch.tensor([s["s"]["t"][0] for s in streams], dtype=torch.float64, device="cuda")
    tc = torch.tensor([s["s"]["t"][1] for s in streams], dtype=torch.float64, device="cuda")
    last = np.stack([s["s"]["ns_true"][0] for s in streams]).copy()
    last[:, 19:22] = [[2e-3, -1e-3, 5e-4]] * B                     # a non-zero delta bias on the last frame
    ns = torch.from_numpy(last).cuda()
    pre = torch.zeros((B, 142), dtype=torch.float64, device="cuda"); cur = torch.zeros((B, 22), dtype=torch.float64, device="cuda")
    pose = torch.zeros((B, 12), dtype=torch.float32, device="cuda")
    fe.imu_predict(imu, tl, tc, ns, pre, cur, pose)
    torch.cuda.synchronize()
    for b, s in enumerate(streams):
        st = dict(s["s"]); st["ns_true"] = [last[b]]
        opre = oracle.preintegrate(st["imu"][1], st["ns_true"][0][10:13], st["ns_true"][0][13:16], st["t"][0], st["t"][1])
        np.testing.assert_allclose(pre[b].cpu().numpy()[:60], opre[:60], rtol=0, atol=1e-12)
        np.testing.assert_allclose(pre[b].cpu().numpy()[60:141], opre[60:141], rtol=1e-9, atol=1e-18)
        assert abs(pre[b, 141].item() - opre[141]) < 1e-14
        ocur = oracle.predict_navstate(st["ns_true"][0], opre, st["gw"])
        np.testing.assert_allclose(cur[b].cpu().numpy(), ocur, rtol=0, atol=1e-11)
        np.testing.assert_allclose(ocur[13:16], last[b, 13:16] + last[b, 19:22], atol=0)
        assert (ocur[16:22] == 0).all()
        Rcw, tcw = cam_pose_from_navstate(ocur, st["cam"])
        np.testing.assert_allclose(pose[b].cpu().numpy()[:9].reshape(3, 3), Rcw, atol=2e-6)
        np.testing.assert_allclose(pose[b].cpu().numpy()[9:], tcw, atol=2e-5)
    # host drop-in
    st = streams[0]["s"]
    hp = viorb_amd.preintegrate(st["imu"][1], st["ns_true"][0][10:13], st["ns_true"][0][13:16], st["t"][0], st["t"][1])
    np.testing.assert_allclose(hp[:60], oracle.preintegrate(st["imu"][1], st["ns_true"][0][10:13], st["ns_true"][0][13:16], st["t"][0], st["t"][1])[:60], atol=1e-12)


@pytest.mark.parametrize("th", [15.0, 30.0, 7.0, 160.0])      # 160: windows of most of the image, far more than the 128 candidates a point's stored list holds
def test_search_by_projection_matches_oracle(torch_cuda, oracle, streams, th):
    torch = torch_cuda
    B, cap = len(streams), 1016
    fe = make_frontend(streams[0], B, cap)
    ck = pad(torch, [s["k1"] for s in streams], cap, viorb_amd.KP_DTYPE)
    cd = pad(torch, [s["d1"] for s in streams], cap, np.uint8, (32,))
    cc = torch.tensor([len(s["k1"]) for s in streams], dtype=torch.int32, device="cuda")
    lk = pad(torch, [s["k0"] for s in streams], cap, viorb_amd.KP_DTYPE)
    ld = pad(torch, [s["d0"] for s in streams], cap, np.uint8, (32,))
    lc = torch.tensor([len(s["k0"]) for s in streams], dtype=torch.int32, device="cuda")
    lf = pad(torch, [s["flags"] for s in streams], cap, np.uint8)
    lp = pad(torch, [s["Pw"] for s in streams], cap, np.float32, (3,))
    poses = []
    for s in streams:
        Rcw, tcw = cam_pose_from_navstate(s["s"]["ns_true"][1], s["s"]["cam"])
        poses.append(np.concatenate([Rcw.ravel(), tcw + (0.01 if th == 30 else 0.0)]).astype(np.float32))
    pose = torch.from_numpy(np.stack(poses)).cuda()
    cs = torch.zeros((B, 3073), dtype=torch.int32, device="cuda"); ci = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    fe.grid(ck.data_ptr(), cc.data_ptr(), B, cs, ci)
    match = torch.full((B, cap), -7, dtype=torch.int32, device="cuda")
    nm = torch.zeros(B, dtype=torch.int32, device="cuda"); status = torch.zeros(B, dtype=torch.int32, device="cuda")
    fe.search_projection(ck.data_ptr(), cd.data_ptr(), cc.data_ptr(), cs, ci, pose, lk.data_ptr(), lc.data_ptr(), lf, lp, ld.data_ptr(),
                         th, B, match, nm, status)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all()
    for b, s in enumerate(streams):
        onm, om = oracle.search_by_projection_frame(s["k1"], s["d1"], BOUNDS, poses[b], s["s"]["cam"][:4], s["tables"]["scale"],
                                                    s["flags"], s["Pw"], s["d0"], s["k0"]["octave"], s["k0"]["angle"], th)
        assert nm[b].item() == onm
        np.testing.assert_array_equal(match[b, :len(om)].cpu().numpy(), om)
        assert (match[b, len(om):].cpu().numpy() == -1).all()
        if th == 15.0:
            assert onm > 250


def _gpu_pose_opt(p, oracle, variant, marg):
    last = p["ns_last"]
    pre = oracle.preintegrate(p["imu"], last[10:13], last[13:16], p["t_last"], p["t_cur"])
    cur0 = oracle.update_ns(last, pre, p["gw"])
    if variant == 0:
        o = oracle.pose_opt_vi_kf(cur0, last, pre, p["gw"], p["cam"], p["obs_cur"], marg=marg)
        g = viorb_amd.PoseOptimization(cur0, last, pre, p["gw"], p["cam"], p["obs_cur"], last_is_keyframe=True, bComputeMarg=marg)
    else:
        o = oracle.pose_opt_vi_frame(cur0, last, p["prior"], p["marg_cov_inv"], pre, p["gw"], p["cam"], p["obs_cur"], p["obs_last"], marg=marg)
        g = viorb_amd.PoseOptimization(cur0, last, pre, p["gw"], p["cam"], p["obs_cur"], p["obs_last"], p["prior"], p["marg_cov_inv"],
                                       last_is_keyframe=False, bComputeMarg=marg)
    return o, g


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_pose_optimisation_matches_oracle(torch_cuda, oracle, variant, seed):
    p = make_vio_problem(seed, n_points=300 if seed < 3 else 600)
    o, g = _gpu_pose_opt(p, oracle, variant, True)
    assert abs(g["final_chi2"] - o["final_chi2"]) <= 1e-5 * abs(o["final_chi2"])          # north-star tolerance
    assert g["n_inliers"] == o["n_inliers"]
    np.testing.assert_array_equal(g["outlier_cur"], o["outlier_cur"])
    if variant == 1:
        np.testing.assert_array_equal(g["outlier_last"], o["outlier_last"])
        np.testing.assert_allclose(g["ns_last"], o["ns_last"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(g["ns"], o["ns"], rtol=0, atol=1e-7)
    assert g["lm_iterations"] == o["lm_iterations"]
    M, Mo = g["marg_cov_inv"], o["marg_cov_inv"]
    np.testing.assert_allclose(M, Mo, rtol=1e-4, atol=1e-6 * np.abs(Mo).max())


def test_pose_optimisation_rounds_that_end_on_a_rejected_trial(torch_cuda, oracle):
    """g2o leaves the errors of a REJECTED last trial on the active edges (optimization_algorithm_levenberg.cpp:143-147 restores the vertices,
    nothing recomputes the errors) and Optimizer.cc:629-634 / :659-664 classify the inliers by that stored chi2, re-evaluating only the edges
    that were outliers. Oracle and device both do exactly that (former deviation 4). A solve restarted from its own optimum
    ends its rounds on ten rejected trials: the oracle reports those rounds (`rejected_rounds`), and flags / inlier counts / LM iteration
    counts / cost still agree — VI solver (both overloads) and the vision-only solver."""
    from viorb_amd.synth import make_se3_problem
    seen = 0
    for seed in (0, 5, 9, 14, 16, 20):
        p = make_vio_problem(seed, n_points=120 + (seed % 4) * 100)
        last = p["ns_last"]
        pre = oracle.preintegrate(p["imu"], last[10:13], last[13:16], p["t_last"], p["t_cur"])
        cur = oracle.update_ns(last, pre, p["gw"])
        for rep in range(4):                                         # every repetition starts where the previous one ended
            o = oracle.pose_opt_vi_kf(cur, last, pre, p["gw"], p["cam"], p["obs_cur"], marg=True)
            g = viorb_amd.PoseOptimization(cur, last, pre, p["gw"], p["cam"], p["obs_cur"], last_is_keyframe=True, bComputeMarg=True)
            seen += o["rejected_rounds"] > 0
            # restarted at the optimum the accept / reject decisions are rounding noise (rho = 0 +- 1e-16 / scale): the iteration COUNT is only
            # comparable for the first solve; the verdicts, the cost and the state must agree every time
            assert g["n_inliers"] == o["n_inliers"] and (rep > 0 or g["lm_iterations"] == o["lm_iterations"]), (seed, rep)
            np.testing.assert_array_equal(g["outlier_cur"], o["outlier_cur"])
            assert abs(g["final_chi2"] - o["final_chi2"]) <= 1e-5 * abs(o["final_chi2"])
            np.testing.assert_allclose(g["ns"], o["ns"], rtol=0, atol=1e-7)
            cur = o["ns"]
    assert seen >= 6, "the restarted solves must exercise rounds that end on a rejected trial"
    seen = 0
    for seed in (1, 2, 3, 14, 15, 19):
        p = make_se3_problem(seed, n_points=200 + 50 * (seed % 5), stereo_frac=[0, 0.6, 1.0][seed % 3])
        intr5 = p["intr5"].astype(np.float32)
        pose = p["pose0"]
        for rep in range(4):
            o = oracle.pose_opt_se3(pose, intr5.astype(np.float64), p["obs7"])
            g = viorb_amd.PoseOptimizationSE3(pose, intr5, p["obs7"])
            seen += o["rejected_rounds"] > 0
            assert g["n_inliers"] == o["n_inliers"] and (rep > 0 or g["lm_iterations"] == o["lm_iterations"]), (seed, rep)
            np.testing.assert_array_equal(g["outlier"], o["outlier"])
            assert abs(g["final_chi2"] - o["final_chi2"]) <= 1e-5 * abs(o["final_chi2"])
            pose = o["pose12"]
    assert seen >= 6


@pytest.mark.parametrize("n_points", [3, 63, 255, 256, 257, 511, 512, 513, 769])
def test_pose_optimisation_edge_counts_around_the_thread_count(torch_cuda, oracle, n_points):
    """The solver's edge loop takes 256 edges per trip, two register sets in turn: counts at and around the trip boundaries."""
    p = make_vio_problem(7, n_points=n_points)
    o, g = _gpu_pose_opt(p, oracle, 1, True)
    assert g["n_inliers"] == o["n_inliers"] and g["lm_iterations"] == o["lm_iterations"]
    assert abs(g["final_chi2"] - o["final_chi2"]) <= 1e-5 * abs(o["final_chi2"]) + 1e-12
    np.testing.assert_array_equal(g["outlier_cur"], o["outlier_cur"])
    np.testing.assert_array_equal(g["outlier_last"], o["outlier_last"])
    np.testing.assert_allclose(g["ns"], o["ns"], rtol=0, atol=1e-7)


@pytest.mark.parametrize("n_points,variant", [(5200, 0), (9000, 1)])
def test_pose_optimisation_more_edges_than_the_searches_hold_keypoints(torch_cuda, oracle, n_points, variant):
    """The host drop-in builds its handle for the number of edges; the keypoint limit of the projection search (LDS plan,
    viorb_frontend_search_capacity) must not apply to a handle that only solves."""
    assert n_points > viorb_amd.lib().viorb_frontend_search_capacity()
    p = make_vio_problem(n_points, n_points=n_points)
    o, g = _gpu_pose_opt(p, oracle, variant, True)
    assert abs(g["final_chi2"] - o["final_chi2"]) <= 1e-5 * abs(o["final_chi2"]) and g["n_inliers"] == o["n_inliers"]
    np.testing.assert_array_equal(g["outlier_cur"], o["outlier_cur"])
    assert g["lm_iterations"] == o["lm_iterations"]
    np.testing.assert_allclose(g["ns"], o["ns"], rtol=0, atol=1e-7)


def test_pose_optimisation_random_sweep(torch_cuda, oracle):
    """Forty random problems per overload: the discrete outcomes (inliers, outlier flags, LM iterations) never differ from the oracle's."""
    for variant in (0, 1):
        for seed in range(200, 240):
            p = make_vio_problem(seed, n_points=150 + 17 * (seed % 30))
            o, g = _gpu_pose_opt(p, oracle, variant, False)
            assert g["n_inliers"] == o["n_inliers"] and g["lm_iterations"] == o["lm_iterations"], (variant, seed)
            np.testing.assert_array_equal(g["outlier_cur"], o["outlier_cur"])
            assert abs(g["final_chi2"] - o["final_chi2"]) <= 1e-5 * abs(o["final_chi2"]) + 1e-12


def test_pose_optimisation_edge_cases(torch_cuda, oracle):
    p = make_vio_problem(5)
    last = p["ns_last"]
    pre = oracle.preintegrate(p["imu"], last[10:13], last[13:16], p["t_last"], p["t_cur"])
    cur0 = oracle.update_ns(last, pre, p["gw"])
    g = viorb_amd.PoseOptimization(cur0, last, pre, p["gw"], p["cam"], p["obs_cur"][:2])       # < 3 correspondences -> 0
    assert g["n_inliers"] == 0
    np.testing.assert_allclose(g["ns"][:10], cur0[:10], atol=1e-15)
    o = oracle.pose_opt_vi_kf(cur0, last, pre, p["gw"], p["cam"], p["obs_cur"][:6])            # < 10 edges: one round
    g = viorb_amd.PoseOptimization(cur0, last, pre, p["gw"], p["cam"], p["obs_cur"][:6])
    assert g["n_inliers"] == o["n_inliers"] and abs(g["final_chi2"] - o["final_chi2"]) <= 1e-5 * abs(o["final_chi2"]) + 1e-12
    bad = p["obs_cur"].copy(); bad[::2, 3:5] += 60.0                                            # half the matches are gross outliers
    o = oracle.pose_opt_vi_kf(cur0, last, pre, p["gw"], p["cam"], bad)
    g = viorb_amd.PoseOptimization(cur0, last, pre, p["gw"], p["cam"], bad)
    assert g["n_inliers"] == o["n_inliers"]
    np.testing.assert_array_equal(g["outlier_cur"], o["outlier_cur"])


def test_batched_pose_opt_and_observation_builder(torch_cuda, oracle, streams):
    """Device-resident chain: build_observations -> pose_opt (Frame/Frame overload) for a batch of streams."""
    torch = torch_cuda
    B, cap = len(streams), 1016
    fe = make_frontend(streams[0], B, cap)
    ck = pad(torch, [s["k1"] for s in streams], cap, viorb_amd.KP_DTYPE)
    cc = torch.tensor([len(s["k1"]) for s in streams], dtype=torch.int32, device="cuda")
    lk = pad(torch, [s["k0"] for s in streams], cap, viorb_amd.KP_DTYPE)
    lc = torch.tensor([len(s["k0"]) for s in streams], dtype=torch.int32, device="cuda")
    lp = pad(torch, [s["Pw"] for s in streams], cap, np.float32, (3,))
    matches, omatches = [], []
    for s in streams:
        Rcw, tcw = cam_pose_from_navstate(s["s"]["ns_true"][1], s["s"]["cam"])
        pose = np.concatenate([Rcw.ravel(), tcw]).astype(np.float32)
        nm, m = oracle.search_by_projection_frame(s["k1"], s["d1"], BOUNDS, pose, s["s"]["cam"][:4], s["tables"]["scale"], s["flags"],
                                                  s["Pw"], s["d0"], s["k0"]["octave"], s["k0"]["angle"], 15.0)
        matches.append(m)
    cm = pad(torch, matches, cap, np.int32) - (pad(torch, [np.ones(len(m), np.int32) for m in matches], cap, np.int32) == 0).int()
    lm = pad(torch, [np.where(s["flags"] & 1, np.arange(len(s["flags"])), -1).astype(np.int32) for s in streams], cap, np.int32) \
        - (pad(torch, [np.ones(len(s["flags"]), np.int32) for s in streams], cap, np.int32) == 0).int()
    obs_c = torch.zeros((B, cap, 6), dtype=torch.float64, device="cuda"); obs_l = torch.zeros_like(obs_c)
    idx_c = torch.zeros((B, cap), dtype=torch.int32, device="cuda"); idx_l = torch.zeros_like(idx_c)
    n_c = torch.zeros(B, dtype=torch.int32, device="cuda"); n_l = torch.zeros_like(n_c)
    fe.build_observations(ck.data_ptr(), cc.data_ptr(), cm, lp, B, obs_c, idx_c, n_c)
    fe.build_observations(lk.data_ptr(), lc.data_ptr(), lm, lp, B, obs_l, idx_l, n_l)
    # NavStates: last = truth (slightly perturbed), cur = IMU prediction
    last_ns = np.stack([s["s"]["ns_true"][0] for s in streams])
    imu = torch.from_numpy(np.stack([s["s"]["imu"][1] for s in streams])).cuda()
    tl = torch.tensor([s["s"]["t"][0] for s in streams], dtype=torch.float64, device="cuda")
    tc = torch.tensor([s["s"]["t"][1] for s in streams], dtype=torch.float64, device="cuda")
    lns = torch.from_numpy(last_ns).cuda()
    pre = torch.zeros((B, 142), dtype=torch.float64, device="cuda"); cur = torch.zeros((B, 22), dtype=torch.float64, device="cuda")
    pose = torch.zeros((B, 12), dtype=torch.float32, device="cuda")
    fe.imu_predict(imu, tl, tc, lns, pre, cur, pose)
    mci = torch.from_numpy(np.stack([np.eye(12) * 1e3] * B)).cuda()
    out = torch.zeros((B, 22), dtype=torch.float64, device="cuda"); outl = torch.zeros_like(out)
    fc = torch.zeros((B, cap), dtype=torch.uint8, device="cuda"); fl = torch.zeros_like(fc)
    marg = torch.zeros((B, 144), dtype=torch.float64, device="cuda"); info = torch.zeros((B, 4), dtype=torch.float64, device="cuda")
    fe.pose_opt(1, True, cur, lns, lns, mci, pre, obs_c, n_c, obs_l, n_l, B, out, outl, fc, fl, marg, info)
    torch.cuda.synchronize()
    for b, s in enumerate(streams):
        m = matches[b]
        sel = np.nonzero(m >= 0)[0]
        want = np.concatenate([s["Pw"][m[sel]].astype(np.float64), np.stack([s["k1"]["x"][sel], s["k1"]["y"][sel]], 1).astype(np.float64),
                               s["tables"]["inv_sigma2"][s["k1"]["octave"][sel]].astype(np.float64)[:, None]], 1)
        assert n_c[b].item() == len(sel)
        np.testing.assert_array_equal(idx_c[b, :len(sel)].cpu().numpy(), sel)
        np.testing.assert_array_equal(obs_c[b, :len(sel)].cpu().numpy(), want)
        ol_np = obs_l[b, :n_l[b].item()].cpu().numpy()
        o = oracle.pose_opt_vi_frame(cur[b].cpu().numpy(), last_ns[b], last_ns[b], np.eye(12) * 1e3, pre[b].cpu().numpy(), s["s"]["gw"],
                                     s["s"]["cam"], want, ol_np, marg=True)
        gi = info[b].cpu().numpy()
        assert int(gi[0]) == o["n_inliers"] and abs(gi[1] - o["final_chi2"]) <= 1e-5 * o["final_chi2"]
        np.testing.assert_array_equal(fc[b, :len(sel)].cpu().numpy(), o["outlier_cur"])
        np.testing.assert_allclose(out[b].cpu().numpy(), o["ns"], atol=1e-7)
        # the optimised pose is close to the ground truth of the synthetic stream
        assert np.linalg.norm(out[b].cpu().numpy()[:3] - s["s"]["ns_true"][1][:3]) < 0.02


def test_host_dropin_search_by_projection(torch_cuda, oracle, streams):
    s = streams[1]
    Rcw, tcw = cam_pose_from_navstate(s["s"]["ns_true"][1], s["s"]["cam"])
    pose = np.concatenate([Rcw.ravel(), tcw]).astype(np.float32)
    m = viorb_amd.ORBmatcher(0.9, True)
    for th in (15.0, 30.0):
        nm, match = m.SearchByProjection(s["k1"], s["d1"], BOUNDS, pose, s["s"]["cam"][:4], s["tables"]["scale"], s["k0"], s["flags"],
                                         s["Pw"], s["d0"], th)
        onm, om = oracle.search_by_projection_frame(s["k1"], s["d1"], BOUNDS, pose, s["s"]["cam"][:4], s["tables"]["scale"], s["flags"],
                                                    s["Pw"], s["d0"], s["k0"]["octave"], s["k0"]["angle"], th)
        assert nm == onm
        np.testing.assert_array_equal(match, om)
    nm, match = m.SearchByProjection(s["k1"][:0], s["d1"][:0], BOUNDS, pose, s["s"]["cam"][:4], s["tables"]["scale"], s["k0"], s["flags"],
                                     s["Pw"], s["d0"], 15.0)
    assert nm == 0 and len(match) == 0


def test_host_dropin_search_when_the_scratch_arena_grows_mid_call(torch_cuda, oracle, streams):
    """A host-buffer drop-in stages its inputs in a page-locked mirror of the calling thread's device arena. A thread whose arena is
    still the initial 4 MiB block and whose frame has more than 16384 keypoints (scratch ~4.75 MB) makes the arena grow in the middle
    of the call, after buffers have been zero-staged in the first block and before their data is put(): the data must still arrive
    (round-3 advisor finding: flush() copied the staged zeros over it and the search returned VIORB_OK with wrong matches).
    Run on a fresh thread (fresh thread-local arena): a small call first, then the large one, both against the oracle."""
    from concurrent.futures import ThreadPoolExecutor
    s = streams[0]
    Rcw, tcw = cam_pose_from_navstate(s["s"]["ns_true"][1], s["s"]["cam"])
    pose = np.concatenate([Rcw.ravel(), tcw]).astype(np.float32)
    rng = np.random.default_rng(3)
    reps = 17000 // len(s["k1"]) + 1
    kbig = np.tile(s["k1"], reps)[:17000].copy()
    kbig["x"] = np.clip(kbig["x"] + rng.uniform(-6, 6, len(kbig)).astype(np.float32), 1, 750)
    kbig["y"] = np.clip(kbig["y"] + rng.uniform(-6, 6, len(kbig)).astype(np.float32), 1, 478)
    dbig = np.tile(s["d1"], (reps, 1))[:17000].copy()
    flip = rng.random(dbig.shape) < 0.15
    dbig[flip] ^= rng.integers(1, 256, flip.sum(), dtype=np.uint8)
    intr, sf = s["s"]["cam"][:4], s["tables"]["scale"]

    def on_a_fresh_thread():
        m = viorb_amd.ORBmatcher(0.9, True)
        small = m.SearchByProjection(s["k1"], s["d1"], BOUNDS, pose, intr, sf, s["k0"], s["flags"], s["Pw"], s["d0"], 15.0)
        big = m.SearchByProjection(kbig, dbig, BOUNDS, pose, intr, sf, s["k0"], s["flags"], s["Pw"], s["d0"], 15.0)
        again = m.SearchByProjection(kbig, dbig, BOUNDS, pose, intr, sf, s["k0"], s["flags"], s["Pw"], s["d0"], 15.0)     # arena already large: the plain path
        return small, big, again
    with ThreadPoolExecutor(max_workers=1) as exr:
        small, big, again = exr.submit(on_a_fresh_thread).result()
    onm, om = oracle.search_by_projection_frame(s["k1"], s["d1"], BOUNDS, pose, intr, sf, s["flags"], s["Pw"], s["d0"], s["k0"]["octave"], s["k0"]["angle"], 15.0)
    assert small[0] == onm; np.testing.assert_array_equal(small[1], om)
    bnm, bm = oracle.search_by_projection_frame(kbig, dbig, BOUNDS, pose, intr, sf, s["flags"], s["Pw"], s["d0"], s["k0"]["octave"], s["k0"]["angle"], 15.0)
    assert bnm > 100
    assert big[0] == bnm; np.testing.assert_array_equal(big[1], bm)
    assert again[0] == bnm; np.testing.assert_array_equal(again[1], bm)


@pytest.mark.parametrize("motion", [0.0, 0.5, -0.5])
def test_host_dropin_search_by_projection_stereo_branch(torch_cuda, oracle, motion):
    """SearchByProjection(CurrentFrame, LastFrame, th, bMono=false): forward / backward octave windows and the mvuRight gate
    (reference src/ORBmatcher.cc:1346-1349, 1385-1410), device vs oracle."""
    from test_oracle_matcher import stereo_scenario
    from viorb_amd.synth import make_vi_stream
    s = make_vi_stream(1, 2)
    ex = oracle.Extractor()
    k0, d0 = ex(s["frames"][0]); k1, d1 = ex(s["frames"][1])
    pair = (s, ex.tables()["scale"], k0, d0, k1, d1)
    pose, last_pose, intr, flags, Pw, mpd, loct, lang, ur, bf, mb = stereo_scenario(pair, motion)
    m = viorb_amd.ORBmatcher(0.9, True)
    for th in (7.0, 14.0):
        nm, match = m.SearchByProjection(k1, d1, BOUNDS, pose, intr, pair[1], k0, flags, Pw, mpd, th, bMono=False, cur_uright=ur, last_pose12=last_pose,
                                         bf=bf, mb=mb)
        onm, om = oracle.search_by_projection_frame_stereo(k1, d1, ur, BOUNDS, pose, last_pose, intr, bf, mb, pair[1], flags, Pw, mpd, loct, lang, th)
        assert nm == onm and nm > 50
        np.testing.assert_array_equal(match, om)


@pytest.mark.parametrize("th,nnratio", [(1.0, 0.8), (5.0, 0.8), (60.0, 0.8)])     # 60: windows with hundreds of candidates (no capacity limit)
def test_search_local_points_matches_oracle(torch_cuda, oracle, th, nnratio):
    """a12: isInFrustum + SearchByProjection(Frame, local map points), two streams in one launch."""
    torch = torch_cuda
    from viorb_amd.synth import make_vi_stream, make_local_map, plane_points_f32
    ex = oracle.Extractor()
    sf = ex.tables()["scale"]
    scenes = []
    for seed in (4, 6):
        s = make_vi_stream(seed, 3)
        feats = [ex(f) for f in s["frames"]]
        pts, descs = [], []
        for j in (0, 1):
            k, d = feats[j]
            Rcw, tcw = cam_pose_from_navstate(s["ns_true"][j], s["cam"])
            Pw = plane_points_f32(np.stack([k["x"], k["y"]], 1), np.concatenate([Rcw.ravel(), tcw]), s["cam"])
            pts.append(make_local_map(k, Pw, s["ns_true"][j], s["cam"], sf)); descs.append(d)
        rng = np.random.default_rng(seed)
        pts_f, pts_desc = np.concatenate(pts), np.concatenate(descs)
        flags = np.full(len(pts_f), 1 | 4, np.uint8)
        flags[rng.random(len(flags)) < 0.05] &= ~np.uint8(1); flags[rng.random(len(flags)) < 0.3] |= 2; flags[rng.random(len(flags)) < 0.1] &= ~np.uint8(4)
        k2, d2 = feats[2]
        owner = (rng.random(len(k2)) < 0.3).astype(np.uint8)
        Rcw, tcw = cam_pose_from_navstate(s["ns_true"][2], s["cam"])
        scenes.append(dict(s=s, k2=k2, d2=d2, pose=np.concatenate([Rcw.ravel(), tcw]).astype(np.float32), pts_f=pts_f, pts_desc=pts_desc,
                           flags=flags, owner=owner))
    B, cap, pcap = len(scenes), 1016, 2100
    st0 = dict(s=scenes[0]["s"], tables=ex.tables())
    fe = make_frontend(st0, B, cap)
    ck = pad(torch, [c["k2"] for c in scenes], cap, viorb_amd.KP_DTYPE); cd = pad(torch, [c["d2"] for c in scenes], cap, np.uint8, (32,))
    cc = torch.tensor([len(c["k2"]) for c in scenes], dtype=torch.int32, device="cuda")
    cs = torch.zeros((B, 3073), dtype=torch.int32, device="cuda"); ci = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    fe.grid(ck.data_ptr(), cc.data_ptr(), B, cs, ci)
    pf = pad(torch, [c["pts_f"] for c in scenes], pcap, np.float32, (8,)); pfl = pad(torch, [c["flags"] for c in scenes], pcap, np.uint8)
    pd = pad(torch, [c["pts_desc"] for c in scenes], pcap, np.uint8, (32,))
    pc = torch.tensor([len(c["pts_f"]) for c in scenes], dtype=torch.int32, device="cuda")
    own = pad(torch, [c["owner"] for c in scenes], cap, np.uint8)
    pose = torch.from_numpy(np.stack([c["pose"] for c in scenes])).cuda()
    match = torch.full((B, cap), -7, dtype=torch.int32, device="cuda"); nm = torch.zeros(B, dtype=torch.int32, device="cuda")
    fr = torch.zeros((B, pcap, 5), dtype=torch.float32, device="cuda"); status = torch.zeros(B, dtype=torch.int32, device="cuda")
    fe.search_local_points(ck.data_ptr(), cd.data_ptr(), cc.data_ptr(), cs, ci, pose, pf, pfl, pd, pc, th, nnratio, own, B, match, nm, fr, status)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all()
    for b, c in enumerate(scenes):
        onm, om, ofr = oracle.search_local_points(c["k2"], c["d2"], BOUNDS, c["pose"], c["s"]["cam"][:4], sf, np.float32(np.log(np.float64(np.float32(1.2)))),
                                                  c["pts_f"], c["flags"], c["pts_desc"], th, nnratio, c["owner"])
        np.testing.assert_array_equal(fr[b, :len(ofr)].cpu().numpy(), ofr)
        assert nm[b].item() == onm and onm > 100
        np.testing.assert_array_equal(match[b, :len(om)].cpu().numpy(), om)
        # the host-buffer drop-in for Tracking.cc:1955 (viorb_search_by_projection_points) gives the same answer
        hn, hm, hfr = viorb_amd.SearchLocalPoints(c["k2"], c["d2"], BOUNDS, c["pose"], c["s"]["cam"][:4], sf, c["pts_f"], c["flags"], c["pts_desc"], th, nnratio,
                                                  c["owner"], want_frustum=True)
        assert hn == onm
        np.testing.assert_array_equal(hm, om); np.testing.assert_array_equal(hfr, ofr)
    n0, m0 = viorb_amd.SearchLocalPoints(scenes[0]["k2"][:0], scenes[0]["d2"][:0], BOUNDS, scenes[0]["pose"], scenes[0]["s"]["cam"][:4], sf,
                                         scenes[0]["pts_f"], scenes[0]["flags"], scenes[0]["pts_desc"])
    assert n0 == 0 and len(m0) == 0


@pytest.mark.parametrize("shape", [(752, 480, 1000), (1241, 376, 2000)])
def test_search_local_points_stereo_gate_matches_oracle(torch_cuda, oracle, shape):
    """a12 with a stereo / RGB-D current frame (reference src/ORBmatcher.cc:91-97, src/Frame.cc:499): a candidate keypoint that has a right
    coordinate is skipped when |mTrackProjXR - mvuRight| exceeds the window radius. EuRoC-sized and KITTI-shaped (1241x376, 2000 features)
    frames; right coordinates from the true depth for 60 % of the keypoints (+- 0.5 px), 5-60 px off for 20 %, none for the rest. Device
    entry and host-buffer drop-in against the oracle: matches, nmatches, the five isInFrustum fields and mTrackProjXR, bit for bit; the gate
    must change the result (fewer matches than the monocular search of the same frame)."""
    torch = torch_cuda
    from viorb_amd.synth import make_vi_stream, make_local_map, plane_points_f32
    w, h, nfeat = shape
    bounds = (0.0, float(w), 0.0, float(h))
    bf = np.float32(386.1448)
    ex = oracle.Extractor(nfeat, 1.2, 8, 20, 7)
    sf = ex.tables()["scale"]
    scenes = []
    for seed in (5, 9):
        s = make_vi_stream(seed, 3, w=w, h=h)
        feats = [ex(f) for f in s["frames"]]
        pts, descs = [], []
        for j in (0, 1):
            k, d = feats[j]
            Rcw, tcw = cam_pose_from_navstate(s["ns_true"][j], s["cam"])
            Pw = plane_points_f32(np.stack([k["x"], k["y"]], 1), np.concatenate([Rcw.ravel(), tcw]), s["cam"])
            pts.append(make_local_map(k, Pw, s["ns_true"][j], s["cam"], sf)); descs.append(d)
        rng = np.random.default_rng(seed + 77)
        pts_f, pts_desc = np.concatenate(pts), np.concatenate(descs)
        flags = np.full(len(pts_f), 1 | 4, np.uint8)
        flags[rng.random(len(flags)) < 0.05] &= ~np.uint8(1); flags[rng.random(len(flags)) < 0.2] |= 2; flags[rng.random(len(flags)) < 0.1] &= ~np.uint8(4)
        k2, d2 = feats[2]
        owner = (rng.random(len(k2)) < 0.2).astype(np.uint8)
        Rcw, tcw = cam_pose_from_navstate(s["ns_true"][2], s["cam"])
        pose = np.concatenate([Rcw.ravel(), tcw]).astype(np.float32)
        # mvuRight of the current frame from the true depth of every keypoint
        Pw2 = plane_points_f32(np.stack([k2["x"], k2["y"]], 1), np.concatenate([Rcw.ravel(), tcw]), s["cam"]).astype(np.float64)
        z = (Pw2 @ Rcw.T + tcw)[:, 2]
        ur = (k2["x"].astype(np.float64) - float(bf) / z + rng.uniform(-0.5, 0.5, len(k2)))
        u = rng.random(len(k2))
        off = u < 0.2
        ur[off] += rng.choice([-1.0, 1.0], off.sum()) * rng.uniform(5, 60, off.sum())
        ur[(u >= 0.2) & (u < 0.4)] = -1.0
        scenes.append(dict(s=s, k2=k2, d2=d2, pose=pose, pts_f=pts_f, pts_desc=pts_desc, flags=flags, owner=owner, ur=ur.astype(np.float32)))
    B = len(scenes)
    cap = max(len(c["k2"]) for c in scenes) + 16
    pcap = max(len(c["pts_f"]) for c in scenes) + 40
    t = ex.tables()
    fe = viorb_amd.Frontend(scenes[0]["s"]["cam"], scenes[0]["s"]["gw"], t["scale"], t["inv_sigma2"], bounds, max_batch=B, cap=cap)
    ck = pad(torch, [c["k2"] for c in scenes], cap, viorb_amd.KP_DTYPE); cd = pad(torch, [c["d2"] for c in scenes], cap, np.uint8, (32,))
    cc = torch.tensor([len(c["k2"]) for c in scenes], dtype=torch.int32, device="cuda")
    cs = torch.zeros((B, 3073), dtype=torch.int32, device="cuda"); ci = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    fe.grid(ck.data_ptr(), cc.data_ptr(), B, cs, ci)
    pf = pad(torch, [c["pts_f"] for c in scenes], pcap, np.float32, (8,)); pfl = pad(torch, [c["flags"] for c in scenes], pcap, np.uint8)
    pd = pad(torch, [c["pts_desc"] for c in scenes], pcap, np.uint8, (32,))
    pc = torch.tensor([len(c["pts_f"]) for c in scenes], dtype=torch.int32, device="cuda")
    own = pad(torch, [c["owner"] for c in scenes], cap, np.uint8); cur_ur = pad(torch, [c["ur"] for c in scenes], cap, np.float32)
    pose = torch.from_numpy(np.stack([c["pose"] for c in scenes])).cuda()
    log_sf = np.float32(np.log(np.float64(np.float32(1.2))))
    for th, nnratio in ((1.0, 0.8), (3.0, 0.8)):
        match = torch.full((B, cap), -7, dtype=torch.int32, device="cuda"); nm = torch.zeros(B, dtype=torch.int32, device="cuda")
        fr = torch.zeros((B, pcap, 5), dtype=torch.float32, device="cuda"); xr = torch.full((B, pcap), -3.0, dtype=torch.float32, device="cuda")
        status = torch.zeros(B, dtype=torch.int32, device="cuda")
        fe.search_local_points(ck.data_ptr(), cd.data_ptr(), cc.data_ptr(), cs, ci, pose, pf, pfl, pd, pc, th, nnratio, own, B, match, nm, fr, status,
                               cur_uright=cur_ur, bf=float(bf), frustum_xr=xr)
        torch.cuda.synchronize()
        assert (status.cpu().numpy() == 0).all()
        for b, c in enumerate(scenes):
            cam4 = c["s"]["cam"][:4]
            onm, om, ofr, oxr = oracle.search_local_points(c["k2"], c["d2"], bounds, c["pose"], cam4, sf, log_sf, c["pts_f"], c["flags"], c["pts_desc"], th, nnratio,
                                                           c["owner"], cur_uright=c["ur"], bf=float(bf))
            mono_nm, mono_m, _ = oracle.search_local_points(c["k2"], c["d2"], bounds, c["pose"], cam4, sf, log_sf, c["pts_f"], c["flags"], c["pts_desc"], th, nnratio,
                                                            c["owner"])
            assert onm > 100 and onm < mono_nm and (om != mono_m).any(), "the right-coordinate gate must bite on this scene"
            np.testing.assert_array_equal(fr[b, :len(ofr)].cpu().numpy(), ofr)
            called = ((c["flags"] & 1) != 0) & ((c["flags"] & 2) == 0)               # isInFrustum runs for these only
            np.testing.assert_array_equal(xr[b, :len(oxr)].cpu().numpy()[called], oxr[called])
            assert nm[b].item() == onm
            np.testing.assert_array_equal(match[b, :len(om)].cpu().numpy(), om)
            # mTrackProjXR is what Frame::isInFrustum stores: u - mbf * invz in float
            inview = ofr[:, 0] != 0
            assert inview.sum() > 100 and (oxr[inview] < ofr[inview, 1]).all()
            hn, hm, hfr, hxr = viorb_amd.SearchLocalPoints(c["k2"], c["d2"], bounds, c["pose"], cam4, sf, c["pts_f"], c["flags"], c["pts_desc"], th, nnratio, c["owner"],
                                                           want_frustum=True, cur_uright=c["ur"], bf=float(bf))
            assert hn == onm
            np.testing.assert_array_equal(hm, om); np.testing.assert_array_equal(hfr, ofr); np.testing.assert_array_equal(hxr, oxr)
    # a frame whose right coordinates are all "none" is the monocular search
    c = scenes[0]
    hn, hm = viorb_amd.SearchLocalPoints(c["k2"], c["d2"], bounds, c["pose"], c["s"]["cam"][:4], sf, c["pts_f"], c["flags"], c["pts_desc"], 1.0, 0.8, c["owner"],
                                         cur_uright=np.full(len(c["k2"]), -1.0, np.float32), bf=float(bf))
    mn, mm = viorb_amd.SearchLocalPoints(c["k2"], c["d2"], bounds, c["pose"], c["s"]["cam"][:4], sf, c["pts_f"], c["flags"], c["pts_desc"], 1.0, 0.8, c["owner"])
    assert hn == mn and (hm == mm).all()


@pytest.mark.parametrize("stereo_frac", [0.0, 0.6, 1.0])
def test_pose_optimisation_se3_matches_oracle(torch_cuda, oracle, stereo_frac):
    """a21: vision-only PoseOptimization(Frame*) with mono and stereo edges."""
    from viorb_amd.synth import make_se3_problem
    for seed in (0, 1, 2):
        p = make_se3_problem(seed, n_points=300 + 150 * seed, stereo_frac=stereo_frac)
        intr5 = p["intr5"].astype(np.float32)
        o = oracle.pose_opt_se3(p["pose0"], intr5.astype(np.float64), p["obs7"])
        g = viorb_amd.PoseOptimizationSE3(p["pose0"], intr5, p["obs7"])
        assert g["n_inliers"] == o["n_inliers"] and g["lm_iterations"] == o["lm_iterations"]
        np.testing.assert_array_equal(g["outlier"], o["outlier"])
        assert abs(g["final_chi2"] - o["final_chi2"]) <= 1e-5 * o["final_chi2"]
        np.testing.assert_allclose(g["pose12"], o["pose12"], rtol=0, atol=2e-6)
    p = make_se3_problem(5)
    g = viorb_amd.PoseOptimizationSE3(p["pose0"], p["intr5"].astype(np.float32), p["obs7"][:2])
    assert g["n_inliers"] == 0
    np.testing.assert_array_equal(g["pose12"], p["pose0"])


def test_search_projection_retry_only_touches_streams_below_the_limit(oracle):
    """TrackWithIMU's 2*th retry as a batched launch: streams with >= retry_below matches keep their first-pass result bit for bit,
    the others get exactly what SearchByProjection(..., 2*th) returns."""
    import torch, ctypes as C
    import viorb_amd
    from viorb_amd.frontend import Frontend
    from viorb_amd.synth import make_periodic_stream, plane_points_f32
    from viorb_amd.capi import KP_DTYPE
    B = 3
    streams = [make_periodic_stream(70 + b, 2) for b in range(B)]
    ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    tab = ex.tables()
    fe = Frontend(streams[0]["cam"], streams[0]["gw"], tab["scale"], tab["inv_sigma2"], max_batch=B, cap=ex.cap)
    dev = torch.device("cuda", 0); cap = ex.cap
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # last frame = frame 0 (features + plane points), current = frame 1 with the true pose as the prediction
    last = [oracle.Extractor(1000)(s["frames"][0]) for s in streams]
    lk = np.zeros((B, cap), KP_DTYPE); ld = np.zeros((B, cap, 32), np.uint8); lc = np.zeros(B, np.int32); lP = np.zeros((B, cap, 3), np.float32); lf = np.zeros((B, cap), np.uint8)
    for b, (k, d) in enumerate(last):
        n = len(k); lk[b, :n] = k; ld[b, :n] = d; lc[b] = n; lf[b, :n] = 1 | 4
        lP[b, :n] = plane_points_f32(np.stack([k["x"], k["y"]], 1), streams[b]["pose_true"][0], streams[b]["cam"])
    lf[1, 40:] = 0                                               # stream 1 keeps 40 map points only: few matches at th = 15
    ex.extract_batch_device(up(np.stack([s["frames"][1] for s in streams])))
    kps, desc, count, _, _ = ex.results_device()
    cs = torch.zeros((B, 64 * 48 + 1), dtype=torch.int32, device=dev); ci = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    fe.grid(kps, count, B, cs, ci)
    pose = up(np.stack([s["pose_true"][1] for s in streams]).astype(np.float32))
    m = torch.zeros((B, cap), dtype=torch.int32, device=dev); nm = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
    tl = [up(x) for x in (lk.view(np.uint8).reshape(B, -1), lc, lf, lP, ld)]
    args = (kps, desc, count, cs, ci, pose, tl[0].data_ptr(), tl[1].data_ptr(), tl[2], tl[3], tl[4].data_ptr())
    fe.search_projection(*args, 15.0, B, m, nm, st)
    torch.cuda.synchronize(); m1, n1 = m.cpu().numpy().copy(), nm.cpu().numpy().copy()
    limit = int(n1[1]) + 1                                       # only stream 1 is below the limit
    assert n1[0] >= limit and n1[2] >= limit
    fe.search_projection(*args, 30.0, B, m, nm, st, retry_below=limit)
    torch.cuda.synchronize(); m2, n2 = m.cpu().numpy(), nm.cpu().numpy()
    assert np.array_equal(m2[0], m1[0]) and np.array_equal(m2[2], m1[2]) and n2[0] == n1[0] and n2[2] == n1[2]
    ck, cd = ex.download(1)
    r_n, r_m = oracle.search_by_projection_frame(ck, cd, (0.0, 752.0, 0.0, 480.0), streams[1]["pose_true"][1].astype(np.float32), streams[1]["cam"][:4],
                                                 tab["scale"], lf[1, :lc[1]], lP[1, :lc[1]], ld[1, :lc[1]], lk[1, :lc[1]]["octave"], lk[1, :lc[1]]["angle"], 30.0)
    assert n2[1] == r_n and np.array_equal(m2[1, :len(ck)], r_m) and (st.cpu().numpy() == 0).all()


@pytest.fixture
def pose_shape():
    """Sets the VI pose solver's launch shape for one test and restores the automatic choice afterwards."""
    L = viorb_amd.lib()
    def set_shape(P, W):
        assert L.viorb_frontend_set_pose_shape(int(P), int(W)) == 0
    yield set_shape
    L.viorb_frontend_set_pose_shape(0, 0)


@pytest.mark.parametrize("shape", [(2, 2), (1, 4), (1, 8), (4, 2), (2, 3)])
def test_pose_solver_shapes_equal_the_oracle(torch_cuda, oracle, pose_shape, shape):
    """k_pose_opt_vi_mp<P, WPP>: the shape a deployment's batch size selects — <2, 2> from two problems per CU (the benchmark's), <1, 4> / <1, 8>
    for small batches — and two more of the instantiated ones. Small test batches would only ever run <1, 8>, so the shape is forced: a batch of
    5 problems of different sizes and BOTH overloads in one launch (per-problem `variant`, uneven lock-step partners, a last workgroup that is
    only partly filled, a problem with < 3 correspondences that returns at once) against the oracle: flags, inlier counts, LM iteration counts
    exact, cost 1e-5, state 1e-7, marginal 1e-4."""
    torch = torch_cuda
    pose_shape(*shape)
    probs = [make_vio_problem(40 + i, n_points=n) for i, n in enumerate((300, 90, 650, 2, 420))]
    variants = [1, 0, 1, 1, 0]
    B, cap = len(probs), 700
    fe = viorb_amd.Frontend(probs[0]["cam"], probs[0]["gw"], np.float32(1.2) ** np.arange(8), 1.0 / (np.float32(1.2) ** np.arange(8)) ** 2, BOUNDS, max_batch=B, cap=cap)
    cur, last, pre, oc, ol = [], [], [], [], []
    for p in probs:
        l = p["ns_last"]; pr = oracle.preintegrate(p["imu"], l[10:13], l[13:16], p["t_last"], p["t_cur"])
        cur.append(oracle.update_ns(l, pr, p["gw"])); last.append(l); pre.append(pr)
        a = np.zeros((cap, 6)); a[:len(p["obs_cur"])] = p["obs_cur"]; oc.append(a)
        b = np.zeros((cap, 6)); b[:len(p["obs_last"])] = p["obs_last"]; ol.append(b)
    up = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dt))).cuda()
    d = dict(cur=up(cur), last=up(last), prior=up([p["prior"] for p in probs]), mci=up([p["marg_cov_inv"].ravel() for p in probs]), pre=up(pre), oc=up(oc), ol=up(ol),
             nc=up([len(p["obs_cur"]) for p in probs], np.int32), nl=up([len(p["obs_last"]) for p in probs], np.int32), var=up(variants, np.uint8), skip=up([0] * B, np.uint8))
    out = dict(ns=torch.zeros((B, 22), dtype=torch.float64, device="cuda"), nsl=torch.zeros((B, 22), dtype=torch.float64, device="cuda"),
               fc=torch.zeros((B, cap), dtype=torch.uint8, device="cuda"), fl=torch.zeros((B, cap), dtype=torch.uint8, device="cuda"),
               marg=torch.zeros((B, 144), dtype=torch.float64, device="cuda"), info=torch.zeros((B, 4), dtype=torch.float64, device="cuda"))
    from viorb_amd.capi import check, ptr
    check(viorb_amd.lib().viorb_frontend_pose_opt_select_device(fe.h, ptr(d["var"]), ptr(d["skip"]), 1, ptr(d["cur"]), ptr(d["last"]), ptr(d["prior"]), ptr(d["mci"]), ptr(d["pre"]),
                                                                ptr(d["oc"]), ptr(d["nc"]), ptr(d["ol"]), ptr(d["nl"]), B, ptr(out["ns"]), ptr(out["nsl"]), ptr(out["fc"]), ptr(out["fl"]),
                                                                ptr(out["marg"]), ptr(out["info"]), None))
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in out.items()}
    for i, (p, var) in enumerate(zip(probs, variants)):
        o = (oracle.pose_opt_vi_frame(cur[i], last[i], p["prior"], p["marg_cov_inv"], pre[i], p["gw"], p["cam"], p["obs_cur"], p["obs_last"], marg=True) if var else
             oracle.pose_opt_vi_kf(cur[i], last[i], pre[i], p["gw"], p["cam"], p["obs_cur"], marg=True))
        n = len(p["obs_cur"])
        assert int(g["info"][i, 0]) == o["n_inliers"] and int(g["info"][i, 2]) == o["lm_iterations"], (shape, i)
        if n < 3:
            np.testing.assert_array_equal(g["ns"][i], cur[i])
            continue
        assert abs(g["info"][i, 1] - o["final_chi2"]) <= 1e-5 * abs(o["final_chi2"]), (shape, i)
        np.testing.assert_array_equal(g["fc"][i, :n], o["outlier_cur"])
        np.testing.assert_allclose(g["ns"][i], o["ns"], rtol=0, atol=1e-7)
        if var:
            np.testing.assert_array_equal(g["fl"][i, :len(p["obs_last"])], o["outlier_last"])
            np.testing.assert_allclose(g["nsl"][i], o["ns_last"], rtol=0, atol=1e-7)
        M = o["marg_cov_inv"]
        np.testing.assert_allclose(g["marg"][i].reshape(12, 12), M, rtol=1e-4, atol=1e-6 * np.abs(M).max())

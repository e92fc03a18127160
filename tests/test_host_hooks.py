"""CPU tests of the product's host-side code (no GPU): the C ABI exports what include/viorb.h
declares, and the host formulations that the kernels mirror agree with the oracle."""
import ctypes as C
import os
import re
import numpy as np
import pytest
import viorb_amd
from viorb_amd import capi
from viorb_amd.extractor import octree_host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "viorb.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(viorb_[a-z0-9_]+)\s*\(", txt)))


def declared_arg_counts():
    """name -> number of parameters of every prototype in include/viorb.h (void = 0)."""
    txt = open(os.path.join(ROOT, "include", "viorb.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(viorb_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_ctypes_signatures_have_the_declared_argument_counts():
    counts = declared_arg_counts()
    assert len(counts) >= 60
    for name, n in counts.items():
        assert name in capi.SIGNATURES, name
        assert len(capi.SIGNATURES[name][1]) == n, "%s: header declares %d parameters, capi.py %d" % (name, n, len(capi.SIGNATURES[name][1]))
    for name in capi.SIGNATURES:
        assert name in counts, "capi.py declares %s, include/viorb.h does not" % name


def test_library_exports_every_declared_symbol():
    L = viorb_amd.lib()
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "include/viorb.h declares %s but libviorb_hip.so does not export it" % n
        assert n in capi.SIGNATURES, "capi.py has no ctypes signature for %s" % n
    assert L.viorb_abi_version() == 2
    assert isinstance(L.viorb_last_error(), bytes)


def test_no_cpu_fallback_without_gpu():
    L = viorb_amd.lib()
    if L.viorb_device_count() > 0:
        pytest.skip("GPU present")
    ex = viorb_amd.ORBextractor()
    with pytest.raises(viorb_amd.ViorbError) as e:
        ex(np.zeros((480, 752), np.uint8))
    assert e.value.code == capi.ERR_NO_DEVICE
    k, d = ex(np.zeros((0, 0), np.uint8))                   # empty image: silent, like the reference
    assert len(k) == 0 and d.shape == (0, 32)


def test_create_argument_errors():
    L = viorb_amd.lib()
    h = C.c_void_p()
    bad = capi.ExtractorParams(1000, 1.2, 0, 20, 7)
    assert L.viorb_extractor_create(C.byref(bad), 1, 0, C.byref(h)) == capi.ERR_INVALID_ARG
    assert b"nlevels" in L.viorb_last_error()
    ok = capi.ExtractorParams(1000, 1.2, 8, 20, 7)
    assert L.viorb_extractor_create(C.byref(ok), 0, 0, C.byref(h)) == capi.ERR_INVALID_ARG


def test_tables_match_oracle(oracle):
    for nf, sf, nl in ((1000, 1.2, 8), (2000, 1.2, 8), (1500, 1.2, 8), (500, 1.5, 5)):
        t = viorb_amd.ORBextractor(nf, sf, nl, 20, 7).tables()
        o = oracle.Extractor(nf, sf, nl, 20, 7).tables()
        for k in ("scale", "inv_scale", "sigma2", "inv_sigma2", "quota"):
            np.testing.assert_array_equal(t[k], o[k])


@pytest.mark.parametrize("seed", range(6))
def test_octree_arrays_equal_list_oracle(oracle, seed):
    """The flat-array quadtree (what k_octree mirrors) returns the same keypoints, in the same order,
    as the std::list restatement of the reference, over many sizes and quotas."""
    rng = np.random.default_rng(seed)
    W, H = [(720, 448), (1209, 344), (595, 368), (178, 102), (1248, 688), (331, 199)][seed]
    for n, N in ((1, 10), (2, 1), (50, 217), (300, 60), (1800, 217), (2500, 434), (1200, 87), (4000, 326), (700, 699)):
        n = min(n, W * H // 4)
        xy = rng.choice(W * H, n, replace=False)
        keys = np.zeros(n, oracle.KP_DTYPE)
        keys["x"], keys["y"] = xy % W, xy // W
        keys["response"] = rng.integers(7, 60 if seed % 2 else 255, n)      # many response ties when narrow
        want = oracle.distribute_octree(keys, 16, 16 + W, 16, 16 + H, N)
        got = octree_host(np.stack([keys["x"], keys["y"], keys["response"]], 1).astype(np.int32), W, H, N)
        np.testing.assert_array_equal(got[:, 0], want["x"].astype(np.int32))
        np.testing.assert_array_equal(got[:, 1], want["y"].astype(np.int32))
        np.testing.assert_array_equal(got[:, 2], want["response"].astype(np.int32))


def test_octree_clustered_points(oracle):
    """Dense clusters force long single-child chains and the largest-first phase with many size ties."""
    rng = np.random.default_rng(42)
    W, H = 720, 448
    pts = set()
    for _ in range(25):
        cx, cy = rng.integers(20, W - 20), rng.integers(20, H - 20)
        for _ in range(80):
            pts.add((int(np.clip(cx + rng.integers(-9, 10), 0, W - 1)), int(np.clip(cy + rng.integers(-9, 10), 0, H - 1))))
    pts = np.array(sorted(pts, key=lambda p: (p[1], p[0])), np.int32)
    keys = np.zeros(len(pts), oracle.KP_DTYPE)
    keys["x"], keys["y"] = pts[:, 0], pts[:, 1]
    keys["response"] = rng.integers(7, 30, len(pts))
    for N in (30, 100, 217, 600):
        want = oracle.distribute_octree(keys, 16, 16 + W, 16, 16 + H, N)
        got = octree_host(np.stack([keys["x"], keys["y"], keys["response"]], 1).astype(np.int32), W, H, N)
        np.testing.assert_array_equal(got, np.stack([want["x"], want["y"], want["response"]], 1).astype(np.int32))


def test_octree_on_real_candidates(oracle):
    from viorb_amd.synth import make_image
    ex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    ex(make_image(3))
    quota = ex.tables()["quota"]
    for l in range(8):
        cand = ex.level_keypoints(l, candidates=True)
        h, w = ex.level(l).shape
        want = ex.level_keypoints(l)
        got = octree_host(np.stack([cand["x"], cand["y"], cand["response"]], 1).astype(np.int32), w - 32, h - 32, int(quota[l]))
        np.testing.assert_array_equal(got[:, 0] + 16, want["x"].astype(np.int32))
        np.testing.assert_array_equal(got[:, 1] + 16, want["y"].astype(np.int32))


def test_device_math_host_build(oracle):
    """orb_math.h compiled for the host: fast_atan2 is bit-identical to the oracle's, sincos_f32 is the
    correctly rounded float of the double-precision value (what the oracle's descriptor uses)."""
    L = viorb_amd.lib()
    rng = np.random.default_rng(5)
    for _ in range(5000):
        y, x = (int(v) for v in rng.integers(-200000, 200000, 2))
        assert L.viorb_debug_fast_atan2(y, x) == oracle.fast_atan2(y, x)
    ang = np.concatenate([rng.uniform(0, 360, 200000), np.arange(0, 360, 0.25), [0, 30, 45, 60, 90, 180, 270, 359.99997]]).astype(np.float32)
    rad = (ang * np.float32(np.pi / np.float32(180.0))).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    want_s = np.sin(rad.astype(np.float64)).astype(np.float32)
    want_c = np.cos(rad.astype(np.float64)).astype(np.float32)
    for i in range(0, len(rad), 7):
        L.viorb_debug_sincos(float(rad[i]), C.byref(s), C.byref(c))
        assert s.value == want_s[i] and c.value == want_c[i]


# ---- vio_core.h (what the solver kernel evaluates) on the host vs the oracle ----------------------------
def _vio_case(oracle, seed):
    from viorb_amd.synth import make_vio_problem
    p = make_vio_problem(seed)
    pre = oracle.preintegrate(p["imu"], p["ns_last"][10:13], p["ns_last"][13:16], p["t_last"], p["t_cur"])
    ni = p["ns_last"].copy(); ni[19:22] = [1e-3, -2e-3, 5e-4]
    nj = oracle.update_ns(p["ns_last"], pre, p["gw"])
    nj = oracle.ns_inc_pvr(nj, np.array([0.01, -0.02, 0.015, 0.03, 0.01, -0.02, 0.01, -0.008, 0.012]))
    return p, pre, ni, nj


@pytest.mark.parametrize("seed", [0, 3])
def test_vio_core_edges_equal_oracle(oracle, seed):
    L = viorb_amd.lib()
    p, pre, ni, nj = _vio_case(oracle, seed)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    e, J = np.zeros(9), np.zeros(9 * 21)
    L.viorb_debug_pvr_edge(P(ni), P(nj), P(ni), P(pre), P(p["gw"]), P(e), P(J))
    oe, Ji, Jj, Jb = oracle.edge_pvr(ni, nj, ni, pre, p["gw"])
    np.testing.assert_allclose(e, oe, rtol=0, atol=1e-12)
    np.testing.assert_allclose(J.reshape(9, 21), np.hstack([Ji, Jj, Jb]), rtol=0, atol=1e-10)
    for k in range(8):
        e2, J12 = np.zeros(2), np.zeros(12)
        ob = np.ascontiguousarray(p["obs_cur"][k])
        L.viorb_debug_proj_edge(P(nj), P(p["cam"]), P(ob), P(e2), P(J12))
        oe2, oJ = oracle.edge_proj(nj, p["cam"], ob)
        np.testing.assert_allclose(e2, oe2, rtol=0, atol=1e-9)
        np.testing.assert_allclose(J12[:6].reshape(2, 3), oJ[:, 0:3], rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(J12[6:].reshape(2, 3), oJ[:, 6:9], rtol=1e-12, atol=1e-10)
    prior = oracle.ns_inc_pvr(ni, np.array([0.02, 0.01, -0.01, 0.02, -0.03, 0.01, 0.004, -0.006, 0.003]))
    e12, J144 = np.zeros(12), np.zeros(144)
    L.viorb_debug_prior_edge(P(ni), P(ni), P(prior), P(e12), P(J144))
    oe12, Jp, Jbb = oracle.edge_prior(ni, ni, prior)
    np.testing.assert_allclose(e12, oe12, rtol=0, atol=1e-12)
    np.testing.assert_allclose(J144.reshape(12, 12), np.hstack([Jp, Jbb]), rtol=0, atol=1e-12)


def test_vio_core_preint_step_and_update_ns_equal_oracle(oracle):
    L = viorb_amd.lib()
    p, pre, ni, nj = _vio_case(oracle, 1)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    small = np.zeros(60); small[6] = small[10] = small[14] = 1
    full = np.zeros(142); full[6] = full[10] = full[14] = 1
    bg, ba = p["ns_last"][10:13], p["ns_last"][13:16]
    for k in range(len(p["imu"])):
        om = np.ascontiguousarray(p["imu"][k, :3] - bg); ac = np.ascontiguousarray(p["imu"][k, 3:6] - ba)
        L.viorb_debug_preint_step(P(small), P(om), P(ac), 0.005)
        full = oracle.preint_update(full, om, ac, 0.005)
    np.testing.assert_allclose(small, full[:60], rtol=0, atol=1e-13)
    out, pose = np.zeros(22), np.zeros(12, np.float32)
    L.viorb_debug_update_ns(P(p["ns_last"]), P(pre), P(p["gw"]), P(p["cam"]), P(out), P(pose))
    want = oracle.update_ns(p["ns_last"], pre, p["gw"])
    np.testing.assert_allclose(out, want, rtol=0, atol=1e-12)
    from viorb_amd.synth import cam_pose_from_navstate
    Rcw, tcw = cam_pose_from_navstate(want, p["cam"])
    np.testing.assert_allclose(pose[:9].reshape(3, 3), Rcw, atol=2e-6)
    np.testing.assert_allclose(pose[9:], tcw, atol=2e-5)


def test_descriptor_distance_host(oracle):
    rng = np.random.default_rng(1)
    for _ in range(100):
        a, b = rng.integers(0, 256, 32, dtype=np.uint8), rng.integers(0, 256, 32, dtype=np.uint8)
        assert viorb_amd.descriptor_distance(a, b) == oracle.descriptor_distance(a, b)


def test_cpp_shim_header_compiles_links_and_runs(tmp_path):
    """viorb_amd/shim/ORBextractor.h (the drop-in for the reference's include/ORBextractor.h) compiles against
    include/viorb.h with stand-in cv:: types, links libviorb_hip.so and behaves like the reference on an empty image."""
    import subprocess
    exe = str(tmp_path / "shim_test")
    lib_dir = os.path.join(ROOT, "viorb_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "viorb_amd", "shim"),
                           "-I", os.path.join(ROOT, "tests", "cpp"), os.path.join(ROOT, "tests", "cpp", "shim_extractor_test.cpp"),
                           "-L", lib_dir, "-lviorb_hip", "-Wl,-rpath," + lib_dir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


def _build_tracking_shim_test(tmp_path):
    import subprocess
    exe = str(tmp_path / "shim_tracking_test")
    lib_dir = os.path.join(ROOT, "viorb_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "viorb_amd", "shim"),
                           "-I", os.path.join(ROOT, "tests", "cpp"), os.path.join(ROOT, "tests", "cpp", "shim_tracking_test.cpp"),
                           "-L", lib_dir, "-lviorb_hip", "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def _build_callsites_shim_test(tmp_path):
    import subprocess
    exe = str(tmp_path / "shim_callsites_test")
    lib_dir = os.path.join(ROOT, "viorb_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "viorb_amd", "shim"),
                           "-I", os.path.join(ROOT, "tests", "cpp"), os.path.join(ROOT, "tests", "cpp", "shim_callsites_test.cpp"),
                           "-L", lib_dir, "-lviorb_hip", "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_cpp_callsite_shims_compile_link_and_refuse_without_a_device(tmp_path):
    """viorb_amd/shim/{ORBmatcher,Frame,Optimizer}_shim.h — the nine call-site templates of INTEGRATION.md (both SearchByProjection forms,
    SearchByBoW, SearchForTriangulation, Fuse, UndistortKeyPoints / ComputeImageBounds, ComputeStereoMatches, ComputeBoW,
    PoseOptimization(Frame*), both local window solves) compile against stand-ins that carry the reference's member names and link
    libviorb_hip.so; without a device every one of them throws (the error is surfaced, nothing falls back to the CPU)."""
    import subprocess
    exe = _build_callsites_shim_test(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


def test_cpp_tracking_shim_compiles_links_and_runs(tmp_path):
    """viorb_amd/shim/viorb_tracking_shim.h (the templates behind Optimizer::PoseOptimization(Frame*, Frame* | KeyFrame*, ...) in
    INTEGRATION.md) compiles against stand-ins that carry the reference's member names, links libviorb_hip.so and, when there is no
    device, throws with the library's error text and leaves the frame untouched — no CPU fallback, no silent "0 inliers"."""
    import subprocess
    exe = _build_tracking_shim_test(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr

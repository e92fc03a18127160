"""-m gpu: the reference-side glue templates of viorb_amd/shim/viorb_tracking_shim.h, driven from a C++ program with stand-in
Frame / NavState / IMUPreintegrator types (tests/cpp/shim_tracking_test.cpp), against the direct C-ABI call on the same problem."""
import os
import subprocess
import numpy as np
import pytest
import viorb_amd
from viorb_amd.synth import make_vio_problem
from test_host_hooks import _build_tracking_shim_test, _build_callsites_shim_test

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [2, 11])
def test_tracking_shim_equals_direct_calls(tmp_path, oracle, seed):
    if viorb_amd.lib().viorb_device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X (and never fall back)")
    p = make_vio_problem(seed, n_points=250)
    last = p["ns_last"]
    pre = oracle.preintegrate(p["imu"], last[10:13], last[13:16], p["t_last"], p["t_cur"])
    cur0 = oracle.update_ns(last, pre, p["gw"])
    # the shim reads map points as float (MapPoint::GetWorldPos is CV_32F) and inverse sigma^2 from the frame's float level table
    def as_frame_sees(obs):
        o = obs.copy()
        o[:, :3] = np.float32(o[:, :3]).astype(np.float64)
        # (float)(1 / pow((double)(float)pow((double)1.2f, l), 2)), as fill_frame() of the C++ test builds mvInvLevelSigma2
        lv64 = np.array([float(np.float32(1.0 / float(np.float32(float(np.float32(1.2)) ** l)) ** 2)) for l in range(8)])
        o[:, 5] = lv64[p["octave"][:len(o)]]
        return o
    oc, ol = as_frame_sees(p["obs_cur"]), as_frame_sees(p["obs_last"])
    cam = p["cam"].copy(); cam[:4] = np.float32(cam[:4]).astype(np.float64)       # Frame::fx, fy, cx, cy are float in the reference
    blob = np.concatenate([cur0, last, p["prior"], p["marg_cov_inv"].ravel(), pre, p["gw"], cam, [len(oc)], oc.ravel(), [len(ol)], ol.ravel()])
    fin, fout = str(tmp_path / "problem.bin"), str(tmp_path / "out.bin")
    blob.astype(np.float64).tofile(fin)
    exe = _build_tracking_shim_test(tmp_path)
    out = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
    r = np.fromfile(fout, np.float64)
    per = 2 + 22 + len(oc) + len(ol) + 144
    assert len(r) == 2 * per
    for k, variant in enumerate((1, 0)):
        blk = r[k * per:(k + 1) * per]
        g = viorb_amd.PoseOptimization(cur0, last, pre, p["gw"], cam, oc, ol if variant else None,
                                       p["prior"] if variant else None, p["marg_cov_inv"] if variant else None,
                                       last_is_keyframe=(variant == 0), bComputeMarg=True)
        assert int(blk[0]) == g["n_inliers"] and g["n_inliers"] > 100
        assert int(blk[1]) == 1                                            # UpdatePoseFromNS called once
        np.testing.assert_array_equal(blk[2:24], g["ns"])
        np.testing.assert_array_equal(blk[24:24 + len(oc)].astype(np.uint8), g["outlier_cur"])
        if variant:
            np.testing.assert_array_equal(blk[24 + len(oc):24 + len(oc) + len(ol)].astype(np.uint8), g["outlier_last"])
        np.testing.assert_array_equal(blk[per - 144:].reshape(12, 12), g["marg_cov_inv"])


def test_callsite_shims_equal_direct_calls(tmp_path):
    """The nine call-site templates (viorb_amd/shim/{ORBmatcher,Frame,Optimizer}_shim.h) driven from C++ with stand-in Frame / KeyFrame /
    MapPoint / Map types: what each writes into the objects equals a direct C-ABI call on independently flattened arrays
    (tests/cpp/shim_callsites_test.cpp)."""
    if viorb_amd.lib().viorb_device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X (and never fall back)")
    exe = _build_callsites_shim_test(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK all nine call-site templates equal the direct C-ABI calls"), out.stdout + out.stderr

"""Row x2 on the device: Frame::UndistortKeyPoints / ComputeImageBounds (reference src/Frame.cc:584-644) through the C ABI against the oracle
restatement of OpenCV 2.4 cvUndistortPoints — bit-exact floats (the kernel runs the same double sequence)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    from viorb_amd.synth import EUROC_K, EUROC_DIST
    K = np.array([EUROC_K["fx"], EUROC_K["fy"], EUROC_K["cx"], EUROC_K["cy"]], np.float32)
    return K, [np.array(EUROC_DIST, np.float32), np.array([-0.3, 0.1, 1e-3, -2e-3, -0.02], np.float32), np.array([0.12, -0.05, 0, 0, 0], np.float32)]


def test_undistort_points_and_bounds_equal_oracle():
    import viorb_amd
    from oracle import binding as ora
    K, dists = _cases()
    rng = np.random.default_rng(9)
    xy = np.concatenate([rng.uniform([-20, -20], [780, 500], (20000, 2)), [[0, 0], [752, 0], [0, 480], [752, 480], [K[2], K[3]]]]).astype(np.float32)
    for D in dists:
        got = viorb_amd.UndistortKeyPoints(xy, K, D)
        assert np.array_equal(got.view(np.uint32), ora.undistort_points(xy, K, D).view(np.uint32))
        for (w, h) in ((752, 480), (1280, 720)):
            assert np.array_equal(viorb_amd.ComputeImageBounds(w, h, K, D), ora.image_bounds(w, h, K, D))
    z = np.zeros(5, np.float32)                                   # mDistCoef(0) == 0: pass-through (Frame.cc:586-590, :637-642)
    assert np.array_equal(viorb_amd.UndistortKeyPoints(xy, K, z), xy)
    assert viorb_amd.ComputeImageBounds(1241, 376, K, z).tolist() == [0.0, 1241.0, 0.0, 376.0]
    assert len(viorb_amd.UndistortKeyPoints(np.zeros((0, 2), np.float32), K, dists[0])) == 0


def test_batched_undistort_keeps_every_other_field():
    """viorb_frontend_undistort_device on extractor output: pt replaced, size / angle / response / octave / class_id untouched, records
    beyond count[b] not written."""
    import torch
    import viorb_amd
    from viorb_amd.synth import make_image, euroc_cam, GRAVITY_CAM_WORLD, EUROC_DIST
    from oracle import binding as ora
    dev = torch.device("cuda", 0)
    imgs = np.stack([make_image(31 + i, 752, 480) for i in range(3)])
    ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=3)
    ex.extract_batch_device(torch.from_numpy(imgs).to(dev))
    torch.cuda.synchronize()
    tab = ex.tables()
    cam = euroc_cam()
    fe = viorb_amd.Frontend(cam, GRAVITY_CAM_WORLD, tab["scale"], tab["inv_sigma2"], max_batch=3, cap=ex.cap, dist_coef=EUROC_DIST)
    kps_ptr, desc_ptr, count_ptr, _, _ = ex.results_device()
    out = torch.full((3, ex.cap, 28), 0xAB, dtype=torch.uint8, device=dev)
    fe.undistort(kps_ptr, count_ptr, 3, out)
    torch.cuda.synchronize()
    K = np.asarray(cam[:4], np.float32); D = np.array(EUROC_DIST, np.float32)
    for b in range(3):
        k, _ = ex.download(b)
        got = out[b].cpu().numpy().reshape(-1).view(viorb_amd.KP_DTYPE)
        n = len(k)
        un = ora.undistort_points(np.stack([k["x"], k["y"]], 1), K, D)
        assert np.array_equal(got["x"][:n].view(np.uint32), un[:, 0].view(np.uint32)) and np.array_equal(got["y"][:n].view(np.uint32), un[:, 1].view(np.uint32))
        for f in ("size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(got[f][:n], k[f]), f
        assert (out[b, n:].cpu().numpy() == 0xAB).all()

"""Oracle restatement of Frame::UndistortKeyPoints / ComputeImageBounds (reference src/Frame.cc:584-644; OpenCV 2.4 cvUndistortPoints) against
independent numpy definitions: the five-iteration fixed point written from the published algorithm, and the forward distortion model."""
import numpy as np
from oracle import binding as ora
from viorb_amd.synth import EUROC_K, EUROC_DIST

K4 = np.array([EUROC_K["fx"], EUROC_K["fy"], EUROC_K["cx"], EUROC_K["cy"]], np.float32)
D5 = np.array(EUROC_DIST, np.float32)


def _numpy_undistort(xy, K4, D5, iters=5):
    """cvUndistortPoints in numpy float64 (elementwise IEEE operations in the published order), float32 out."""
    fx, fy, cx, cy = [np.float64(v) for v in K4]
    k1, k2, p1, p2, k3 = [np.float64(v) for v in D5]
    ifx, ify = 1.0 / fx, 1.0 / fy
    x = (xy[:, 0].astype(np.float64) - cx) * ifx
    y = (xy[:, 1].astype(np.float64) - cy) * ify
    x0, y0 = x.copy(), y.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return np.stack([(fx * x + cx).astype(np.float32), (fy * y + cy).astype(np.float32)], 1)


def _distort(xy_un, K4, D5):
    """Forward radial-tangential model (OpenCV's projectPoints convention) applied to undistorted pixels."""
    fx, fy, cx, cy = [np.float64(v) for v in K4]
    k1, k2, p1, p2, k3 = [np.float64(v) for v in D5]
    x = (xy_un[:, 0].astype(np.float64) - cx) / fx; y = (xy_un[:, 1].astype(np.float64) - cy) / fy
    r2 = x * x + y * y
    c = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * c + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * c + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([xd * fx + cx, yd * fy + cy], 1)


def test_undistort_equals_the_numpy_fixed_point_bit_for_bit():
    rng = np.random.default_rng(3)
    xy = np.concatenate([rng.uniform([0, 0], [752, 480], (5000, 2)), [[0, 0], [752, 0], [0, 480], [752, 480], [367.215, 248.375]]]).astype(np.float32)
    for D in (D5, np.array([-0.3, 0.1, 1e-3, -2e-3, -0.02], np.float32), np.array([0.12, -0.05, 0, 0, 0], np.float32)):
        got = ora.undistort_points(xy, K4, D)
        assert np.array_equal(got.view(np.uint32), _numpy_undistort(xy, K4, D).view(np.uint32))


def test_redistorting_returns_the_original_pixel():
    """Definitional check: the forward model applied to the result gives the input back — to 1e-4 px where five iterations have
    converged (normalised radius below 0.35, the inner half of the EuRoC image), within half a pixel everywhere (the fixed point is cut off after five steps)."""
    v, u = np.mgrid[0:480:7, 0:752:7]
    xy = np.stack([u.ravel(), v.ravel()], 1).astype(np.float32)
    un = ora.undistort_points(xy, K4, D5)
    err = np.abs(_distort(un, K4, D5) - xy).max(1)
    r = np.hypot((xy[:, 0] - K4[2]) / K4[0], (xy[:, 1] - K4[3]) / K4[1])
    assert err[r < 0.35].max() < 1e-4, err[r < 0.35].max()
    assert err.max() < 0.5, err.max()
    # the principal point is a fixed point; barrel distortion (k1 < 0) pushes every other pixel outwards
    c = ora.undistort_points(np.array([[K4[2], K4[3]]], np.float32), K4, D5)
    assert np.allclose(c, [[K4[2], K4[3]]], atol=1e-4)
    assert (np.hypot(un[:, 0] - K4[2], un[:, 1] - K4[3]) >= np.hypot(xy[:, 0] - K4[2], xy[:, 1] - K4[3]) - 1e-3).all()


def test_image_bounds():
    b = ora.image_bounds(752, 480, K4, D5)
    corners = ora.undistort_points(np.array([[0, 0], [752, 0], [0, 480], [752, 480]], np.float32), K4, D5)
    assert b[0] == min(corners[0, 0], corners[2, 0]) and b[1] == max(corners[1, 0], corners[3, 0])
    assert b[2] == min(corners[0, 1], corners[1, 1]) and b[3] == max(corners[2, 1], corners[3, 1])
    assert b[0] < 0 and b[2] < 0 and b[1] > 752 and b[3] > 480                  # barrel distortion: the undistorted frame is larger
    # mDistCoef(0) == 0: the image rectangle, and keypoints pass through unchanged (Frame.cc:586-590, :637-642)
    z = np.zeros(5, np.float32)
    assert ora.image_bounds(1241, 376, K4, z).tolist() == [0.0, 1241.0, 0.0, 376.0]


def test_twin_tracks_a_distorted_stream():
    """The oracle twin with the EuRoC coefficients on a stream rendered through the same lens: keypoints are undistorted ahead of the grid
    (several pixels towards the border), the bounds are the undistorted corners, and frame-to-frame tracking holds."""
    from viorb_amd.synth import make_periodic_stream
    from oracle.harness import OracleTracker
    s = make_periodic_stream(11, 3, dist=EUROC_DIST)
    mci = np.eye(12) * 1e3
    tw = OracleTracker(s["cam"], s["gw"], track_local_map=False, dist_coef=EUROC_DIST)
    assert tw.bounds[0] < -100 and tw.bounds[1] > 850
    tw.bootstrap(s["frames"][0], s["pose_true"][0], s["t"][0], s["ns_true"][0], mci)
    raw, _ = tw.ex(s["frames"][0])
    moved = np.hypot(tw.last_kps["x"] - raw["x"], tw.last_kps["y"] - raw["y"])
    assert moved.mean() > 3 and moved.max() > 30 and np.array_equal(tw.last_kps["octave"], raw["octave"])
    r = tw.step(s["frames"][1], s["imu"][1], s["t"][1], s["pose_true"][1])
    assert r["state"] == 0 and r["n_inliers"] > 400, (r["nmatches"], r.get("n_inliers"))
    assert np.linalg.norm(r["final_ns"][:3] - s["ns_true"][1][:3]) < 0.02

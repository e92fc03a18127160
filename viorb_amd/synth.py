"""Seeded synthetic inputs for tests and bench: no dataset is available, so images
are band-limited noise plus random filled rectangles/discs, and consecutive frames of a stream are
the same scene under a small similarity warp. numpy/scipy only; no GPU, no oracle."""
import numpy as np
from scipy import ndimage

EUROC_K = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375)   # reference Examples/ROS/ORB_VIO/launch/euroc.yaml
EUROC_DIST = (-0.28340811, 0.07395907, 0.00019359, 1.762e-05, 0.0)   # Camera.k1 k2 p1 p2 (euroc.yaml:64-67), k3 = 0


def make_image(seed, w=752, h=480, n_shapes=None):
    """u8 [h, w] image: Gaussian-filtered white noise (sigma 3 px, +-40 around 128) + 400..1500
    random filled rectangles/discs with uniform grey levels."""
    rng = np.random.Generator(np.random.PCG64(seed))
    noise = ndimage.gaussian_filter(rng.standard_normal((h, w)), 3.0)
    noise = noise / (np.abs(noise).max() + 1e-12) * 40.0
    img = 128.0 + noise
    if n_shapes is None:
        n_shapes = int(rng.integers(400, 1500))
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_shapes):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        g = rng.uniform(20, 235)
        if rng.random() < 0.5:
            hw, hh = rng.uniform(3, 40), rng.uniform(3, 40)
            x0, x1 = int(max(cx - hw, 0)), int(min(cx + hw, w))
            y0, y1 = int(max(cy - hh, 0)), int(min(cy + hh, h))
            img[y0:y1, x0:x1] = g
        else:
            r = rng.uniform(3, 30)
            x0, x1 = int(max(cx - r, 0)), int(min(cx + r + 1, w))
            y0, y1 = int(max(cy - r, 0)), int(min(cy + r + 1, h))
            m = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2 <= r * r
            img[y0:y1, x0:x1][m] = g
    img += rng.normal(0, 1.5, size=img.shape)          # sensor-like noise so flat areas are not exact ties
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def warp_image(img, dx, dy, roll_deg, seed=0):
    """Same scene moved by (dx, dy) px and rolled by roll_deg about the image centre (bilinear)."""
    h, w = img.shape
    a = np.deg2rad(roll_deg)
    c, s = np.cos(a), np.sin(a)
    R = np.array([[c, -s], [s, c]])                    # output (y,x) -> input (y,x) rotation
    centre = np.array([h / 2.0, w / 2.0])
    offset = centre - R @ centre - np.array([dy, dx])
    out = ndimage.affine_transform(img.astype(np.float32), R, offset=offset, order=1, mode="reflect")
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    out += rng.normal(0, 1.0, size=out.shape)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def make_stream(seed, n_frames, w=752, h=480):
    """A short synthetic camera stream: frame k = base scene warped by a smooth small motion."""
    base = make_image(seed, w, h)
    rng = np.random.Generator(np.random.PCG64(seed + 104729))
    frames, motions = [], []
    dx = dy = roll = 0.0
    for k in range(n_frames):
        frames.append(base if k == 0 else warp_image(base, dx, dy, roll, seed=seed * 1000 + k))
        motions.append((dx, dy, roll))
        dx += rng.uniform(-4, 4); dy += rng.uniform(-3, 3); roll += rng.uniform(-0.7, 0.7)
        dx, dy, roll = np.clip(dx, -10, 10), np.clip(dy, -10, 10), np.clip(roll, -3, 3)
    return frames, motions


# ==================================================================================================
# Synthetic visual-inertial problems: one "last frame -> current frame" step.
# Flat layouts (shared with include/viorb.h):
#   navstate[22] = P3 V3 q4(x,y,z,w) bg3 ba3 dbg3 dba3 ; cam[16] = fx fy cx cy Rbc9 Pbc3
#   obs[n,6] = Pw3 u v invSigma2 ; imu[n,7] = gyro3 acc3 t
# ==================================================================================================
EUROC_TBC = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
                      [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                      [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
                      [0.0, 0.0, 0.0, 1.0]])       # reference Examples/ROS/ORB_VIO/launch/euroc.yaml Camera.Tbc
GRAVITY_W = np.array([0.0, 0.0, -9.81])


def _rotvec_to_R(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _R_to_quat(R):
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R).as_quat()           # x, y, z, w
    return q if q[3] >= 0 else -q


def euroc_cam():
    R = EUROC_TBC[:3, :3]
    u, _, vt = np.linalg.svd(R)                     # yaml values are 12-digit: re-orthonormalise
    R = u @ vt
    return np.concatenate([[EUROC_K["fx"], EUROC_K["fy"], EUROC_K["cx"], EUROC_K["cy"]], R.ravel(), EUROC_TBC[:3, 3]])


def navstate(P, V, R, bg=(0, 0, 0), ba=(0, 0, 0)):
    return np.concatenate([P, V, _R_to_quat(R), bg, ba, np.zeros(3), np.zeros(3)]).astype(np.float64)


def make_vio_problem(seed, n_points=300, outlier_frac=0.05, n_imu=10, w=752, h=480, pix_sigma=1.0):
    """One tracking step with ground truth: last frame at t0, current frame at t0 + n_imu*5 ms."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = euroc_cam()
    fx, fy, cx, cy = cam[:4]
    Rbc, Pbc = cam[4:13].reshape(3, 3), cam[13:16]
    dt = 0.005
    T = n_imu * dt
    R0 = _rotvec_to_R(rng.normal(0, 0.4, 3))
    P0 = rng.normal(0, 1.0, 3)
    V0 = rng.normal(0, 0.5, 3)
    omega = rng.normal(0, 0.1, 3)                   # body rate, constant over the step
    a_w = rng.normal(0, 0.5, 3)                     # world acceleration, constant over the step
    bg = rng.normal(0, 0.002, 3)
    ba = rng.normal(0, 0.02, 3)
    t0 = 100.0 + seed * 0.05
    ts = t0 + dt * (np.arange(n_imu) + 0.3)         # IMU stamps are not aligned with the frame stamps
    imu = np.zeros((n_imu, 7))
    for k, t in enumerate(ts):
        Rt = R0 @ _rotvec_to_R(omega * (t - t0))
        imu[k, :3] = omega + bg + rng.normal(0, 1e-3, 3)
        imu[k, 3:6] = Rt.T @ (a_w - GRAVITY_W) + ba + rng.normal(0, 1e-2, 3)
        imu[k, 6] = t
    t1 = t0 + T
    R1 = R0 @ _rotvec_to_R(omega * T)
    P1 = P0 + V0 * T + 0.5 * a_w * T * T
    V1 = V0 + a_w * T

    def project(Rwb, Pwb, Pw):
        Pc = (Rbc.T @ (Rwb.T @ (Pw - Pwb).T)).T - Rbc.T @ Pbc
        return np.stack([fx * Pc[:, 0] / Pc[:, 2] + cx, fy * Pc[:, 1] / Pc[:, 2] + cy], 1), Pc[:, 2]

    # points seen from the current camera, then kept if also inside the last image
    uv = np.stack([rng.uniform(20, w - 20, 4 * n_points), rng.uniform(20, h - 20, 4 * n_points)], 1)
    z = rng.uniform(2, 10, len(uv))
    Pc = np.stack([(uv[:, 0] - cx) / fx * z, (uv[:, 1] - cy) / fy * z, z], 1)
    Pw = (R1 @ (Rbc @ (Pc + Rbc.T @ Pbc).T)).T + P1
    uv0, z0 = project(R0, P0, Pw)
    ok = (z0 > 0.5) & (uv0[:, 0] > 20) & (uv0[:, 0] < w - 20) & (uv0[:, 1] > 20) & (uv0[:, 1] < h - 20)
    Pw, uv, uv0 = Pw[ok][:n_points], uv[ok][:n_points], uv0[ok][:n_points]
    n = len(Pw)
    octave = rng.integers(0, 8, n)
    sig = (1.2 ** octave) * pix_sigma
    inv_sigma2 = (1.0 / (np.float32(1.2) ** octave).astype(np.float32) ** 2).astype(np.float64)

    def observe(uvt, salt):
        r = np.random.Generator(np.random.PCG64(seed * 7 + salt))
        o = uvt + r.normal(0, 1, uvt.shape) * sig[:, None]
        bad = r.random(n) < outlier_frac
        o[bad] += r.choice([-1, 1], (bad.sum(), 2)) * r.uniform(15, 25, (bad.sum(), 2))
        return np.float32(o).astype(np.float64), bad                       # keypoints are float32 in the reference

    oc, bad_c = observe(uv, 1)
    ol, bad_l = observe(uv0, 2)
    obs_cur = np.concatenate([Pw, oc, inv_sigma2[:, None]], 1)
    obs_last = np.concatenate([Pw, ol, inv_sigma2[:, None]], 1)
    ns_last_true = navstate(P0, V0, R0, bg, ba)
    ns_cur_true = navstate(P1, V1, R1, bg, ba)
    # what tracking would hold: last state slightly off, bias estimate slightly off
    ns_last = navstate(P0 + rng.normal(0, 0.01, 3), V0 + rng.normal(0, 0.02, 3), R0 @ _rotvec_to_R(rng.normal(0, 0.003, 3)),
                       bg + rng.normal(0, 2e-4, 3), ba + rng.normal(0, 5e-3, 3))
    prior_info = np.diag(np.concatenate([np.full(3, 1e4), np.full(3, 1e3), np.full(3, 1e5), np.full(3, 1e3)]))
    return dict(cam=cam, gw=GRAVITY_W.copy(), imu=imu, t_last=t0, t_cur=t1, ns_last=ns_last, ns_last_true=ns_last_true,
                ns_cur_true=ns_cur_true, obs_cur=obs_cur, obs_last=obs_last, outlier_cur_true=bad_c, outlier_last_true=bad_l,
                prior=ns_last.copy(), marg_cov_inv=prior_info, octave=octave)


# ==================================================================================================
# A full synthetic visual-inertial stream: a textured plane z = Z0 in the world frame (= the frame of
# the reference camera that sees `base` exactly), a smooth 6-DoF body trajectory, perspective-correct
# rendered frames, and IMU samples consistent with the trajectory.
# ==================================================================================================
PLANE_Z0 = 5.0
GRAVITY_CAM_WORLD = np.array([0.0, 9.81, 0.0])       # world = reference camera frame, y points down


_RAY_CACHE = {}


def pixel_rays(w, h, cam4, dist=None):
    """Normalised ray (x, y, 1) of every pixel of a w x h image for the pinhole intrinsics cam4 and, when given, the lens distortion
    (the inverse model run to convergence); cached per camera — every frame of a stream shares it."""
    key = (w, h, tuple(float(v) for v in cam4), None if dist is None else tuple(float(v) for v in dist))
    r = _RAY_CACHE.get(key)
    if r is None:
        fx, fy, cx, cy = cam4
        v, u = np.mgrid[0:h, 0:w].astype(np.float64)
        xn, yn = (u - cx) / fx, (v - cy) / fy
        if dist is not None and dist[0] != 0:
            xn, yn = undistort_normalized(xn, yn, dist)
        if len(_RAY_CACHE) > 8:
            _RAY_CACHE.clear()
        r = _RAY_CACHE[key] = (xn, yn)
    return r


def undistort_normalized(xd, yd, dist, iters=20):
    """Inverse of the radial-tangential model by fixed-point iteration run to convergence (the TRUE camera of the synthetic world;
    the reference's cv::undistortPoints stops after five iterations)."""
    k1, k2, p1, p2, k3 = [float(v) for v in dist]
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        ic = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x); dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (xd - dx) * ic; y = (yd - dy) * ic
    return x, y


def render_view(base, cam, Rcw, tcw, seed=0, noise=1.0, dist=None):
    """Image of the plane z = PLANE_Z0 (textured with `base` as seen from Tcw = I) from pose (Rcw, tcw). dist = k1 k2 p1 p2 k3: the
    camera has that lens distortion (pixel -> ray through the inverse model), as the EuRoC camera of the reference's settings file."""
    h, w = base.shape
    fx, fy, cx, cy = cam[:4]
    xn, yn = pixel_rays(w, h, cam[:4], dist)
    d = np.stack([xn, yn, np.ones_like(xn)], -1) @ Rcw          # Rcw^T d, row-vector form
    O = -Rcw.T @ tcw
    s = (PLANE_Z0 - O[2]) / d[..., 2]
    X = O[0] + s * d[..., 0]
    Y = O[1] + s * d[..., 1]
    ub = fx * X / PLANE_Z0 + cx
    vb = fy * Y / PLANE_Z0 + cy
    out = ndimage.map_coordinates(base.astype(np.float32), [vb, ub], order=1, mode="reflect")
    if noise > 0:
        out = out + np.random.Generator(np.random.PCG64(seed + 31337)).normal(0, noise, out.shape)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def cam_pose_from_navstate(ns, cam):
    """(Rcw, tcw) in float64 from a NavState and Tbc — Frame::UpdatePoseFromNS, reference src/Frame.cc:88-105."""
    from scipy.spatial.transform import Rotation
    Rwb = Rotation.from_quat(ns[6:10]).as_matrix()
    Rbc, Pbc = cam[4:13].reshape(3, 3), cam[13:16]
    Rcw = (Rwb @ Rbc).T
    Pwc = Rwb @ Pbc + ns[:3]
    return Rcw, -Rcw @ Pwc


def backproject_to_plane(uv, ns, cam):
    """World points on the plane z = PLANE_Z0 seen at pixels uv [n,2] from NavState ns."""
    Rcw, tcw = cam_pose_from_navstate(ns, cam)
    fx, fy, cx, cy = cam[:4]
    d = np.stack([(uv[:, 0] - cx) / fx, (uv[:, 1] - cy) / fy, np.ones(len(uv))], 1) @ Rcw
    O = -Rcw.T @ tcw
    s = (PLANE_Z0 - O[2]) / d[:, 2]
    return O[None, :] + s[:, None] * d


def make_vi_stream(seed, n_frames, w=752, h=480, n_imu=10, imu_dt=0.005):
    """Returns dict(frames [n,h,w] u8, ns_true [n,22], imu list of [n_imu,7] per interval (imu[k] spans
    frame k-1 -> k, imu[0] is None), t [n], cam[16], gw[3])."""
    rng = np.random.Generator(np.random.PCG64(seed + 424243))
    cam = euroc_cam()
    Rbc, Pbc = cam[4:13].reshape(3, 3), cam[13:16]
    base = make_image(seed, w, h)
    # body pose such that the camera starts at Tcw = I:  Rwb = Rbc^T, Pwb = -Rwb Pbc
    R = Rbc.T.copy(); P = -R @ Pbc
    V = rng.normal(0, 0.3, 3) * np.array([1, 1, 0.2])
    bg, ba = rng.normal(0, 0.002, 3), rng.normal(0, 0.02, 3)
    T = n_imu * imu_dt
    t = 50.0 + seed
    frames, states, imus, ts = [], [], [None], []
    for k in range(n_frames):
        ns = navstate(P, V, R, bg, ba)
        Rcw, tcw = cam_pose_from_navstate(ns, cam)
        frames.append(base.copy() if k == 0 else render_view(base, cam, Rcw, tcw, seed=seed * 1009 + k))
        states.append(ns); ts.append(t)
        if k == n_frames - 1:
            break
        omega = rng.normal(0, 0.12, 3)
        a_w = rng.normal(0, 0.8, 3) * np.array([1, 1, 0.3]) - 0.8 * V      # keeps the speed bounded
        stamps = t + imu_dt * (np.arange(n_imu) + 0.3)
        imu = np.zeros((n_imu, 7))
        for j, tj in enumerate(stamps):
            Rt = R @ _rotvec_to_R(omega * (tj - t))
            imu[j, :3] = omega + bg + rng.normal(0, 1e-3, 3)
            imu[j, 3:6] = Rt.T @ (a_w - GRAVITY_CAM_WORLD) + ba + rng.normal(0, 1e-2, 3)
            imu[j, 6] = tj
        imus.append(imu)
        P = P + V * T + 0.5 * a_w * T * T
        V = V + a_w * T
        R = R @ _rotvec_to_R(omega * T)
        t = t + T
    return dict(frames=np.stack(frames), ns_true=np.stack(states), imu=imus, t=np.array(ts), cam=cam,
                gw=GRAVITY_CAM_WORLD.copy(), base=base)


def make_periodic_stream(seed, n_frames=8, w=752, h=480, n_imu=10, imu_dt=0.005, nfeat_hint=None, dist=None):
    """A closed-loop mono-inertial stream: poses, velocities and images are periodic with period
    n_frames * n_imu * imu_dt, so a tracker can run over it for any number of steps (frame k of the run is
    frame k % n_frames of the stream; time keeps increasing). IMU samples come from the analytic trajectory.
    Returns dict(frames [n,h,w], ns_true [n,22], pose_true [n,12] (Rcw,tcw double), imu [n,n_imu,7] with
    imu[j] covering frame j-1 -> j (j = 0: frame n-1 -> n == 0) and stamps relative to the period start,
    t [n] frame stamps in [0, T), period T, cam, gw)."""
    rng = np.random.Generator(np.random.PCG64(seed + 987001))
    cam = euroc_cam()
    Rbc, Pbc = cam[4:13].reshape(3, 3), cam[13:16]
    base = make_image(seed, w, h)
    dtf = n_imu * imu_dt
    T = n_frames * dtf
    wp = 2 * np.pi / T
    A = rng.uniform(0.02, 0.05, 3) * np.array([1, 1, 0.4]); phP = rng.uniform(0, 2 * np.pi, 3)
    Th = np.deg2rad(rng.uniform(0.5, 1.5, 3)); phR = rng.uniform(0, 2 * np.pi, 3)
    bg, ba = rng.normal(0, 0.002, 3), rng.normal(0, 0.02, 3)
    R0 = Rbc.T.copy()                                    # camera starts near Tcw = I
    P0 = -R0 @ Pbc

    def traj(t):
        P = P0 + A * np.sin(wp * t + phP)
        V = A * wp * np.cos(wp * t + phP)
        Acc = -A * wp * wp * np.sin(wp * t + phP)
        th = Th * np.sin(wp * t + phR); thd = Th * wp * np.cos(wp * t + phR)
        R = R0 @ _rotvec_to_R(th)
        n = np.linalg.norm(th)
        if n < 1e-9:
            Jr = np.eye(3)
        else:
            k = th / n; K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            Jr = np.eye(3) - (1 - np.cos(n)) / n * K + (1 - np.sin(n) / n) * K @ K
        return P, V, Acc, R, Jr @ thd

    frames, states, poses, imus, ts = [], [], [], [], []
    for j in range(n_frames):
        t = j * dtf
        P, V, _, R, _ = traj(t)
        ns = navstate(P, V, R, bg, ba)
        Rcw, tcw = cam_pose_from_navstate(ns, cam)
        frames.append(render_view(base, cam, Rcw, tcw, seed=seed * 1013 + j, dist=dist))
        states.append(ns); poses.append(np.concatenate([Rcw.ravel(), tcw])); ts.append(t)
        # IMU between frame j-1 and j (j = 0 closes the loop: stamps in (T - dtf, T))
        tstart = (j - 1) * dtf if j > 0 else T - dtf
        imu = np.zeros((n_imu, 7))
        for q in range(n_imu):
            tq = tstart + imu_dt * (q + 0.3)
            _, _, Acc, Rq, om = traj(tq)
            imu[q, :3] = om + bg + rng.normal(0, 1e-3, 3)
            imu[q, 3:6] = Rq.T @ (Acc - GRAVITY_CAM_WORLD) + ba + rng.normal(0, 1e-2, 3)
            imu[q, 6] = tq
        imus.append(imu)
    return dict(frames=np.stack(frames), ns_true=np.stack(states), pose_true=np.stack(poses), imu=np.stack(imus), t=np.array(ts),
                period=T, frame_dt=dtf, cam=cam, gw=GRAVITY_CAM_WORLD.copy(), base=base)


def plane_points_f32(kps_xy, pose12_true, cam):
    """numpy twin of viorb_synth_plane_points_device (same FP64 operation order, then float32)."""
    T = np.asarray(pose12_true, np.float64)
    fx, fy, cx, cy = cam[:4]
    dx = (kps_xy[:, 0].astype(np.float64) - cx) / fx
    dy = (kps_xy[:, 1].astype(np.float64) - cy) / fy
    rx = T[0] * dx + T[3] * dy + T[6]; ry = T[1] * dx + T[4] * dy + T[7]; rz = T[2] * dx + T[5] * dy + T[8]
    ox = -(T[0] * T[9] + T[3] * T[10] + T[6] * T[11]); oy = -(T[1] * T[9] + T[4] * T[10] + T[7] * T[11])
    oz = -(T[2] * T[9] + T[5] * T[10] + T[8] * T[11])
    s = (PLANE_Z0 - oz) / rz
    return np.stack([ox + s * rx, oy + s * ry, oz + s * rz], 1).astype(np.float32)


def local_points_f32(octaves, pose12_true, Pw, scale_factors):
    """numpy twin of viorb_synth_local_points_device (same FP64 operation order, one rounding to float32):
    pts_f [n,8] = Pw3 normal3 minDist maxDist of points created from one frame with true pose pose12_true."""
    T = np.asarray(pose12_true, np.float64)
    ox = -(T[0] * T[9] + T[3] * T[10] + T[6] * T[11]); oy = -(T[1] * T[9] + T[4] * T[10] + T[7] * T[11])
    oz = -(T[2] * T[9] + T[5] * T[10] + T[8] * T[11])
    P = np.asarray(Pw, np.float32).astype(np.float64)
    dx, dy, dz = P[:, 0] - ox, P[:, 1] - oy, P[:, 2] - oz
    dist = np.sqrt(dx * dx + dy * dy + dz * dz)
    sf = np.asarray(scale_factors, np.float32).astype(np.float64)
    maxd = dist * sf[np.asarray(octaves)]
    mind = maxd / sf[len(sf) - 1]
    return np.concatenate([np.asarray(Pw, np.float32), np.stack([dx / dist, dy / dist, dz / dist, mind, maxd], 1).astype(np.float32)], 1)


def make_local_map(kps, Pw, ns_ref, cam, scale_factors):
    """Local map points from the keypoints of a reference key frame (MapPoint::UpdateNormalAndDepth, reference
    src/MapPoint.cc:339-378): normal = unit viewing ray, mfMaxDistance = dist * scale[octave],
    mfMinDistance = mfMaxDistance / scale[nLevels-1]. Returns pts_f [n,8] float32 = Pw3 normal3 minDist maxDist."""
    Rcw, tcw = cam_pose_from_navstate(ns_ref, cam)
    Ow = -Rcw.T @ tcw
    PO = Pw.astype(np.float64) - Ow
    dist = np.linalg.norm(PO, axis=1)
    normal = PO / dist[:, None]
    sf = np.asarray(scale_factors, np.float64)
    maxd = dist * sf[kps["octave"]]
    mind = maxd / sf[-1]
    return np.concatenate([Pw.astype(np.float64), normal, mind[:, None], maxd[:, None]], 1).astype(np.float32)


def make_se3_problem(seed, n_points=300, stereo_frac=0.0, outlier_frac=0.05, bf=386.1448, w=752, h=480):
    """A vision-only pose problem (Optimizer::PoseOptimization(Frame*)): points in front of a camera with pose
    (Rcw, tcw), float32 observations, an initial pose a few cm / tenths of a degree off. obs7 = Xw3 u v ur invSigma2."""
    rng = np.random.Generator(np.random.PCG64(seed + 5150))
    fx, fy, cx, cy = EUROC_K["fx"], EUROC_K["fy"], EUROC_K["cx"], EUROC_K["cy"]
    Rcw = _rotvec_to_R(rng.normal(0, 0.3, 3)); tcw = rng.normal(0, 0.5, 3)
    uv = np.stack([rng.uniform(20, w - 20, n_points), rng.uniform(20, h - 20, n_points)], 1)
    z = rng.uniform(2, 10, n_points)
    Pc = np.stack([(uv[:, 0] - cx) / fx * z, (uv[:, 1] - cy) / fy * z, z], 1)
    Xw = (Rcw.T @ (Pc - tcw).T).T
    octave = rng.integers(0, 8, n_points)
    sig = 1.2 ** octave
    inv_s2 = (1.0 / (np.float32(1.2) ** octave).astype(np.float32) ** 2).astype(np.float64)
    o = uv + rng.normal(0, 1, uv.shape) * sig[:, None]
    ur = o[:, 0] - bf / z + rng.normal(0, 1, n_points) * sig
    bad = rng.random(n_points) < outlier_frac
    o[bad] += rng.choice([-1, 1], (bad.sum(), 2)) * rng.uniform(15, 25, (bad.sum(), 2))
    is_stereo = rng.random(n_points) < stereo_frac
    ur = np.where(is_stereo, ur, -1.0)
    obs7 = np.concatenate([np.float32(Xw).astype(np.float64), np.float32(o).astype(np.float64), np.float32(ur).astype(np.float64)[:, None],
                           inv_s2[:, None]], 1)
    R0 = _rotvec_to_R(rng.normal(0, 0.004, 3)) @ Rcw
    t0 = tcw + rng.normal(0, 0.03, 3)
    pose0 = np.concatenate([R0.ravel(), t0]).astype(np.float32)
    return dict(pose0=pose0, pose_true=np.concatenate([Rcw.ravel(), tcw]), obs7=obs7, intr5=np.array([fx, fy, cx, cy, bf]), outlier_true=bad)


KITTI_K = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448)    # reference Examples/Stereo/KITTI00-02.yaml


def make_stereo_pair(seed, w=1241, h=376, fx=718.856, bf=386.1448, zmin=4.0, zmax=40.0):
    """Rectified stereo pair with a smooth known depth field: right(x, y) = left(x + d(x, y), y), d = bf / Z.
    Returns (left, right, disparity_of_right_pixels [h, w])."""
    rng = np.random.Generator(np.random.PCG64(seed + 777))
    left = make_image(seed, w, h)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    z = zmin + (zmax - zmin) * (0.5 + 0.5 * np.sin(xx / w * 2.1 + rng.uniform(0, 3)) * np.cos(yy / h * 1.3 + rng.uniform(0, 3)))
    d = bf / z
    right = ndimage.map_coordinates(left.astype(np.float32), [yy, xx + d], order=1, mode="reflect")
    right += np.random.Generator(np.random.PCG64(seed + 778)).normal(0, 1.0, right.shape)
    return left, np.clip(np.rint(right), 0, 255).astype(np.uint8), d


def make_local_ba_problem(seed, W=20, n_points=2000, n_fixed_extra=3, outlier_frac=0.05, pix_sigma=2.0, kf_dt=0.15, imu_dt=0.005,
                          w=752, h=480):
    """A LocalBundleAdjustmentNavState problem: W local key frames on a smooth trajectory, the previous
    key frame and a few older covisible key frames fixed, n_points points each seen by 3..8 key frames, pix_sigma-px
    Gaussian noise, outlier_frac gross outliers at ~20 px. Returns flat arrays (layouts of include/viorb.h) + truth."""
    rng = np.random.Generator(np.random.PCG64(seed + 20202))
    cam = euroc_cam()
    fx, fy, cx, cy = cam[:4]; Rbc, Pbc = cam[4:13].reshape(3, 3), cam[13:16]
    nkf = W + 1 + n_fixed_extra                      # chronological: extras, prev, local window
    n_imu = int(round(kf_dt / imu_dt))
    R = Rbc.T.copy(); Pw = -R @ Pbc; V = rng.normal(0, 0.4, 3) * np.array([1, 1, 0.3])
    bg, ba = rng.normal(0, 0.002, 3), rng.normal(0, 0.02, 3)
    states, preints_imu, t = [], [None], 10.0
    from scipy.spatial.transform import Rotation
    for k in range(nkf):
        states.append(navstate(Pw, V, R, bg, ba))
        if k == nkf - 1:
            break
        omega = rng.normal(0, 0.15, 3); a_w = rng.normal(0, 0.6, 3) * np.array([1, 1, 0.3]) - 0.5 * V
        imu = np.zeros((n_imu, 7))
        for j in range(n_imu):
            tj = t + imu_dt * (j + 0.3)
            Rt = R @ _rotvec_to_R(omega * (tj - t))
            imu[j, :3] = omega + bg + rng.normal(0, 1e-3, 3); imu[j, 3:6] = Rt.T @ (a_w - GRAVITY_CAM_WORLD) + ba + rng.normal(0, 1e-2, 3); imu[j, 6] = tj
        preints_imu.append((imu, t, t + kf_dt))
        Pw = Pw + V * kf_dt + 0.5 * a_w * kf_dt ** 2; V = V + a_w * kf_dt; R = R @ _rotvec_to_R(omega * kf_dt); t += kf_dt
    states = np.stack(states)
    # order for the solver: local window first (chronological), then prev, then extras
    chrono_local = list(range(n_fixed_extra + 1, nkf)); prev_c = n_fixed_extra; extra_c = list(range(n_fixed_extra))
    order = chrono_local + [prev_c] + extra_c
    kfs_true = states[order]
    prev_kf = W
    # points: seen from a random local key frame at depth 2..10 m, then observed by every key frame that sees them
    pts, edges_i, edges_o = [], [], []
    poses = [cam_pose_from_navstate(s, cam) for s in kfs_true]
    sf = np.float32(1.2) ** np.arange(8)
    while len(pts) < n_points:
        k0 = int(rng.integers(0, W))
        u0, v0, z = rng.uniform(30, w - 30), rng.uniform(30, h - 30), rng.uniform(2, 10)
        Rcw, tcw = poses[k0]
        X = Rcw.T @ (np.array([(u0 - cx) / fx * z, (v0 - cy) / fy * z, z]) - tcw)
        vis = []
        for k, (Rk, tk) in enumerate(poses):
            Pc = Rk @ X + tk
            if Pc[2] < 0.5: continue
            uu, vv = fx * Pc[0] / Pc[2] + cx, fy * Pc[1] / Pc[2] + cy
            if 20 < uu < w - 20 and 20 < vv < h - 20: vis.append((k, uu, vv))
        if len(vis) < 3: continue
        nobs = int(min(len(vis), rng.integers(3, 9)))
        sel = sorted(rng.choice(len(vis), nobs, replace=False))
        pid = len(pts); pts.append(X)
        for si in sel:
            k, uu, vv = vis[si]
            octv = int(rng.integers(0, 8)); sg = pix_sigma * float(sf[octv]) / 2.0
            ou, ov = uu + rng.normal(0, sg), vv + rng.normal(0, sg)
            if rng.random() < outlier_frac: ou += rng.choice([-1, 1]) * rng.uniform(15, 25); ov += rng.choice([-1, 1]) * rng.uniform(15, 25)
            edges_i.append((pid, k)); edges_o.append((np.float32(ou), np.float32(ov), 1.0 / float(np.float32(sf[octv]) ** 2)))
    pts = np.array(pts)
    # IMU pre-integrations of the local key frames (predecessor = prev for the first one), with the predecessor's bias
    from . import synth as _self  # noqa: F401
    imu_list = [preints_imu[c] for c in chrono_local]          # interval ending at chrono index c
    # initial estimates
    kfs0 = kfs_true.copy()
    for i in range(W):
        kfs0[i, :3] += rng.normal(0, 0.02, 3); kfs0[i, 3:6] += rng.normal(0, 0.05, 3)
        q = Rotation.from_quat(kfs0[i, 6:10]) * Rotation.from_rotvec(rng.normal(0, 0.005, 3))
        qq = q.as_quat(); kfs0[i, 6:10] = qq if qq[3] >= 0 else -qq
    pts0 = pts + rng.normal(0, 0.05, pts.shape)
    return dict(kfs=kfs0, kfs_true=kfs_true, n_local=W, prev_kf=prev_kf, imu=imu_list, points=np.float32(pts0).astype(np.float64), points_true=pts,
                edge_idx=np.array(edges_i, np.int32), edge_obs=np.array(edges_o, np.float64), gw=GRAVITY_CAM_WORLD.copy(), cam=cam)


def make_vocabulary(seed, k=10, L=6, flip_bits=(96, 64, 40, 24, 14, 8, 5, 3)):
    """A synthetic DBoW2-shaped ORB vocabulary (the reference's ORBvoc.txt is not distributable with the repo): a complete
    k-ary tree of depth L over 256-bit descriptors, each child = its parent with flip_bits[level] random bits flipped
    (root children are uniform random), node ids in breadth-first order shuffled within a level so that child ids are not
    contiguous (as in a k-means-built file), leaves numbered as words in id order with TF-IDF-like weights in (0, 8] and
    about 0.2 % stopped words (weight 0). Flat layout = include/viorb.h `viorb_vocabulary`."""
    rng = np.random.Generator(np.random.PCG64(seed + 777))
    n_level = [k ** l for l in range(L + 1)]
    n_nodes = sum(n_level)
    base = np.cumsum([0] + n_level)
    desc = np.zeros((n_nodes, 32), np.uint8)
    ids = [np.arange(base[l], base[l + 1]) for l in range(L + 1)]
    for l in range(1, L + 1):
        ids[l] = base[l] + rng.permutation(n_level[l])          # node id of the j-th node (in parent-major order) of level l
    child_start = np.zeros(n_nodes + 1, np.int64)
    child_ids = np.zeros(n_nodes - 1, np.int32)
    counts = np.zeros(n_nodes, np.int64)
    for l in range(L):
        counts[ids[l]] = k
    child_start[1:] = np.cumsum(counts)
    for l in range(L):
        par = ids[l]                                            # parents in order j
        ch = ids[l + 1].reshape(n_level[l], k)                  # children of parent j
        pos = child_start[par][:, None] + np.arange(k)[None, :]
        child_ids[pos.ravel()] = ch.ravel()
        if l == 0:
            d = rng.integers(0, 256, (k, 32), dtype=np.uint8)
        else:
            pd = np.repeat(desc[par], k, axis=0)
            nflip = flip_bits[min(l, len(flip_bits) - 1)]
            bits = rng.integers(0, 256, (len(pd), nflip))
            mask = np.zeros((len(pd), 32), np.uint8)
            np.bitwise_xor.at(mask, (np.arange(len(pd))[:, None].repeat(nflip, 1).ravel(), (bits >> 3).ravel()), (1 << (bits & 7)).astype(np.uint8).ravel())
            d = pd ^ mask
        desc[ch.ravel()] = d
    word_id = np.full(n_nodes, -1, np.int32)
    leaves = np.sort(ids[L])
    word_id[leaves] = np.arange(len(leaves), dtype=np.int32)
    weight = np.zeros(n_nodes)
    wl = rng.uniform(0.05, 8.0, len(leaves)); wl[rng.random(len(leaves)) < 0.002] = 0.0
    weight[leaves] = wl
    return dict(k=k, L=L, child_start=child_start.astype(np.int32), child_ids=child_ids, desc=desc, word_id=word_id, weight=weight)


def descriptors_near_words(seed, voc, n, noise_bits=6):
    """n descriptors, each a random vocabulary leaf with noise_bits random bits flipped (so the tree descent is non-trivial
    but features of one scene cluster in a few nodes)."""
    rng = np.random.Generator(np.random.PCG64(seed + 4242))
    leaves = np.nonzero(voc["word_id"] >= 0)[0]
    d = voc["desc"][rng.choice(leaves, n)].copy()
    for _ in range(noise_bits):
        b = rng.integers(0, 256, n)
        d[np.arange(n), b >> 3] ^= (1 << (b & 7)).astype(np.uint8)
    return d


KP_NP = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"), ("octave", "i4"), ("class_id", "i4")])   # == viorb_keypoint / cv::KeyPoint


def make_two_view_problem(seed, n1=900, n2=950, n_common=500, stereo_frac=0.0, w=752, h=480):
    """Two key frames observing a 3-D point cloud, for SearchForTriangulation / Fuse: keypoints (float32 pixel positions with octave
    and angle), descriptors (noisy copies for common points), vocabulary-like node ids (common points share a node), has_point
    flags, the fundamental matrix F12 (x1^T F12 x2 = 0, as ORB-SLAM's ComputeF12), poses and camera centre. numpy only."""
    rng = np.random.Generator(np.random.PCG64(seed + 31337))
    fx, fy, cx, cy = [np.float32(v) for v in (EUROC_K["fx"], EUROC_K["fy"], EUROC_K["cx"], EUROC_K["cy"])]
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    R1 = _rotvec_to_R(rng.normal(0, 0.05, 3)); t1 = rng.normal(0, 0.05, 3)
    R2 = _rotvec_to_R(rng.normal(0, 0.08, 3)) @ R1; t2 = t1 + np.array([0.35, 0.05, 0.02]) + rng.normal(0, 0.02, 3)
    def project(R, t, X):
        Pc = (R @ X.T).T + t
        return np.stack([fx * Pc[:, 0] / Pc[:, 2] + cx, fy * Pc[:, 1] / Pc[:, 2] + cy], 1), Pc[:, 2]
    # common 3-D points in front of both cameras
    X = np.stack([rng.uniform(-3, 3, 4 * n_common), rng.uniform(-2, 2, 4 * n_common), rng.uniform(3, 12, 4 * n_common)], 1)
    uv1, z1 = project(R1, t1, X); uv2, z2 = project(R2, t2, X)
    ok = (z1 > 0.5) & (z2 > 0.5) & (uv1[:, 0] > 20) & (uv1[:, 0] < w - 20) & (uv1[:, 1] > 20) & (uv1[:, 1] < h - 20) & \
         (uv2[:, 0] > 20) & (uv2[:, 0] < w - 20) & (uv2[:, 1] > 20) & (uv2[:, 1] < h - 20)
    X, uv1, uv2 = X[ok][:n_common], uv1[ok][:n_common], uv2[ok][:n_common]
    nc = len(X)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    def frame(n, uvc, salt):
        r = np.random.Generator(np.random.PCG64(seed * 13 + salt))
        k = np.zeros(n, KP_NP)
        k["x"][:nc] = uvc[:, 0] + r.normal(0, 0.7, nc); k["y"][:nc] = uvc[:, 1] + r.normal(0, 0.7, nc)
        k["x"][nc:] = r.uniform(20, w - 20, n - nc); k["y"][nc:] = r.uniform(20, h - 20, n - nc)
        k["octave"] = r.integers(0, 8, n); k["angle"] = r.uniform(0, 360, n); k["size"] = 31 * sf[k["octave"]]; k["class_id"] = -1
        return k
    k1 = frame(n1, uv1, 1); k2 = frame(n2, uv2, 2)
    k2["octave"][:nc] = np.clip(k1["octave"][:nc] + rng.integers(-1, 2, nc), 0, 7)
    k2["angle"][:nc] = (k1["angle"][:nc] + rng.choice([8.0, 8.0, 8.0, 200.0], nc) + rng.normal(0, 3, nc)) % 360
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8); d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    d2[:nc] = d1[:nc]
    for _ in range(14):
        b = rng.integers(0, 256, nc); d2[np.arange(nc), b >> 3] ^= (1 << (b & 7)).astype(np.uint8)
    node1 = rng.integers(0, 60, n1).astype(np.int32); node2 = rng.integers(0, 60, n2).astype(np.int32)
    node2[:nc] = node1[:nc]
    node1[rng.random(n1) < 0.02] = -1
    hp1 = (rng.random(n1) < 0.35).astype(np.uint8); hp2 = (rng.random(n2) < 0.35).astype(np.uint8)
    ur1 = np.where(rng.random(n1) < stereo_frac, k1["x"] - 20.0, -1.0).astype(np.float32)
    ur2 = np.where(rng.random(n2) < stereo_frac, k2["x"] - 20.0, -1.0).astype(np.float32)
    p1 = rng.permutation(n1); p2 = rng.permutation(n2)
    inv2 = np.empty(n2, np.int64); inv2[p2] = np.arange(n2)
    truth = np.full(n1, -1, np.int64); truth[:nc] = np.arange(nc)
    truth = np.where(truth >= 0, inv2[np.maximum(truth, 0)], -1)[p1]
    k1, d1, node1, hp1, ur1 = k1[p1], d1[p1], node1[p1], hp1[p1], ur1[p1]
    k2, d2, node2, hp2, ur2 = k2[p2], d2[p2], node2[p2], hp2[p2], ur2[p2]
    # F12 = K^-T [t12]x R12 K^-1 with R12 = R1 R2^T, t12 = -R1 R2^T t2 + t1 (LocalMapping::ComputeF12), float32 like the cv::Mat
    R12 = R1 @ R2.T; t12 = -R12 @ t2 + t1
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    F12 = (np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)).astype(np.float32)
    Cw1 = (-R1.T @ t1).astype(np.float32)
    pose2 = np.concatenate([R2.ravel(), t2]).astype(np.float32); pose1 = np.concatenate([R1.ravel(), t1]).astype(np.float32)
    return dict(k1=k1, d1=d1, node1=node1, hp1=hp1, ur1=ur1, k2=k2, d2=d2, node2=node2, hp2=hp2, ur2=ur2, F12=F12, Cw1=Cw1, pose1=pose1,
                pose2=pose2, intr4=np.array([fx, fy, cx, cy], np.float32), sf=sf, level_sigma2=(sf * sf).astype(np.float32),
                inv_level_sigma2=(np.float32(1) / (sf * sf)).astype(np.float32), truth12=truth, X=X)


def make_local_ba_se3_problem(seed, W=8, n_fixed=3, n_points=600, stereo_frac=0.5, outlier_frac=0.05, pix_sigma=1.5, w=1241, h=376):
    """A vision-only LocalBundleAdjustment problem (KITTI-shaped camera): W free key frames + n_fixed fixed ones on a forward-moving
    trajectory, points seen by 3..7 key frames, a mix of mono and stereo observations, Gaussian pixel noise and gross outliers.
    Returns flat arrays (kfs [NK,7] = qx qy qz qw tx ty tz of Tcw, free first) + truth."""
    from scipy.spatial.transform import Rotation
    rng = np.random.Generator(np.random.PCG64(seed + 9090))
    fx = fy = 718.856; cx, cy, bf = 607.1928, 185.2157, 386.1448
    nk = W + n_fixed
    poses = []
    p = np.zeros(3); yaw = 0.0
    for k in range(nk):
        Rwc = Rotation.from_euler("y", yaw).as_matrix() @ _rotvec_to_R(rng.normal(0, 0.01, 3))
        poses.append((Rwc.T, -Rwc.T @ p))                                     # Tcw
        p = p + Rwc @ np.array([rng.normal(0, 0.05), rng.normal(0, 0.02), 0.8 + rng.normal(0, 0.1)]); yaw += rng.normal(0, 0.03)
    order = list(range(n_fixed, nk)) + list(range(n_fixed))                   # free (recent) key frames first, then the fixed (older) ones
    poses = [poses[i] for i in order]
    pts, ei, eo = [], [], []
    sf = np.float32(1.2) ** np.arange(8)
    while len(pts) < n_points:
        k0 = int(rng.integers(0, nk)); Rk, tk = poses[k0]
        z = rng.uniform(4, 40); Xc = np.array([(rng.uniform(20, w - 20) - cx) / fx * z, (rng.uniform(20, h - 20) - cy) / fy * z, z])
        X = Rk.T @ (Xc - tk)
        vis = []
        for k, (R, t) in enumerate(poses):
            Pc = R @ X + t
            if Pc[2] < 1.0: continue
            u, v = fx * Pc[0] / Pc[2] + cx, fy * Pc[1] / Pc[2] + cy
            if 10 < u < w - 10 and 10 < v < h - 10: vis.append((k, u, v, Pc[2]))
        if len(vis) < 3: continue
        sel = sorted(rng.choice(len(vis), int(min(len(vis), rng.integers(3, 8))), replace=False))
        pid = len(pts); pts.append(X)
        for si in sel:
            k, u, v, zc = vis[si]
            octv = int(rng.integers(0, 8)); sg = pix_sigma * float(sf[octv]) / 1.5
            ou, ov = u + rng.normal(0, sg), v + rng.normal(0, sg)
            stereo = rng.random() < stereo_frac and zc < 35
            our = (u - bf / zc + rng.normal(0, sg)) if stereo else -1.0
            if rng.random() < outlier_frac: ou += rng.choice([-1, 1]) * rng.uniform(12, 25)
            ei.append((pid, k)); eo.append((np.float32(ou), np.float32(ov), np.float32(our) if stereo else -1.0, 1.0 / float(np.float32(sf[octv]) ** 2)))
    pts = np.array(pts)
    def to7(R, t):
        q = Rotation.from_matrix(R).as_quat(); q = q if q[3] >= 0 else -q
        return np.concatenate([q, t])
    kfs_true = np.stack([to7(R, t) for R, t in poses])
    kfs = kfs_true.copy()
    for i in range(W):                                                        # perturb the free key frames
        R, t = poses[i]
        Rp = _rotvec_to_R(rng.normal(0, 0.01, 3)) @ R
        kfs[i] = to7(Rp, t + rng.normal(0, 0.05, 3))
    points0 = pts + rng.normal(0, 0.1, pts.shape)
    return dict(kfs=kfs, kfs_true=kfs_true, n_local=W, points=points0, points_true=pts, edge_idx=np.array(ei, np.int32), edge_obs=np.array(eo, np.float64),
                intr5=np.array([fx, fy, cx, cy, bf]))

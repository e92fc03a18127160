"""Seeded synthetic inputs for tests and bench: no dataset is available, so images
are band-limited noise plus random filled rectangles/discs, and consecutive frames of a stream are
the same scene under a small similarity warp. numpy/scipy only; no GPU, no oracle."""
import numpy as np
from scipy import ndimage

EUROC_K = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375)   # reference Examples/ROS/ORB_VIO/launch/euroc.yaml


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

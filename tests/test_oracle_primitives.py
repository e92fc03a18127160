"""Pins the CPU oracle's OpenCV-2.4 primitive restatements (oracle/cvprim.cpp) with independent
definitional checks written from scratch in numpy (the reference holds no golden
vectors for this path, so these checks + the constant tables are what pins the oracle)."""
import numpy as np
import pytest
from viorb_amd.synth import make_image

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
          (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def is_corner(img, x, y, t):
    """FAST-9-16 by the literal definition: 9 contiguous circle pixels all brighter than v+t or
    all darker than v-t."""
    v = int(img[y, x])
    ring = [int(img[y + dy, x + dx]) for dx, dy in CIRCLE]
    for sign in (1, -1):
        flags = [(sign * (p - v)) > t for p in ring]
        ff = flags + flags
        run = 0
        for f in ff:
            run = run + 1 if f else 0
            if run >= 9:
                return True
    return False


def corner_score(img, x, y):
    """Largest threshold at which the pixel is still a corner (definition of cv::cornerScore)."""
    lo, hi = -1, 255
    while hi - lo > 1:                     # is_corner is monotone in t
        mid = (lo + hi) // 2
        if is_corner(img, x, y, mid):
            lo = mid
        else:
            hi = mid
    return lo


def fast_definitional(img, t):
    h, w = img.shape
    score = np.zeros((h, w), np.int32)
    corner = np.zeros((h, w), bool)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if is_corner(img, x, y, t):
                corner[y, x] = True
                score[y, x] = corner_score(img, x, y)
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if corner[y, x]:
                nb = score[y - 1:y + 2, x - 1:x + 2].copy()
                nb[1, 1] = -1
                if (score[y, x] > nb).all():
                    out.append((x, y, score[y, x]))
    return np.array(out, np.int32).reshape(-1, 3)


@pytest.mark.parametrize("seed,t", [(0, 20), (1, 7), (2, 20), (3, 7)])
def test_fast_matches_definition(oracle, seed, t):
    img = make_image(seed, 96, 64, n_shapes=40)
    x0, y0, x1, y1 = 5, 4, 5 + 43, 4 + 41             # an off-origin sub-image like a FAST cell
    got = oracle.fast(img, x0, y0, x1, y1, t)
    want = fast_definitional(img[y0:y1, x0:x1], t)
    assert len(want) > 0
    np.testing.assert_array_equal(got, want)


def test_fast_tiny_cells_are_empty(oracle):
    img = make_image(5, 64, 64, n_shapes=30)
    assert len(oracle.fast(img, 0, 0, 6, 40, 7)) == 0      # < 7 columns: no interior pixel
    assert len(oracle.fast(img, 0, 0, 40, 6, 7)) == 0


def test_resize_close_to_float_bilinear(oracle):
    src = make_image(7, 200, 120, n_shapes=60)
    dw, dh = 167, 100
    got = oracle.resize_linear(src, dw, dh).astype(np.float64)
    sx = (np.arange(dw) + 0.5) * (200 / dw) - 0.5
    sy = (np.arange(dh) + 0.5) * (120 / dh) - 0.5
    x0 = np.clip(np.floor(sx).astype(int), 0, 199); x1 = np.clip(x0 + 1, 0, 199); fx = np.clip(sx - x0, 0, 1)
    y0 = np.clip(np.floor(sy).astype(int), 0, 119); y1 = np.clip(y0 + 1, 0, 119); fy = np.clip(sy - y0, 0, 1)
    s = src.astype(np.float64)
    top = s[y0][:, x0] * (1 - fx) + s[y0][:, x1] * fx
    bot = s[y1][:, x0] * (1 - fx) + s[y1][:, x1] * fx
    want = top * (1 - fy)[:, None] + bot * fy[:, None]
    assert np.abs(got - want).max() <= 1.0            # 11-bit fixed point vs exact bilinear
    assert np.abs(got - want).mean() < 0.35


def test_resize_constant_and_identity(oracle):
    c = np.full((50, 70), 137, np.uint8)
    assert (oracle.resize_linear(c, 58, 42) == 137).all()
    src = make_image(8, 64, 48, n_shapes=20)
    np.testing.assert_array_equal(oracle.resize_linear(src, 64, 48), src)      # scale 1 is exact


def test_gaussian_kernel_and_blur(oracle):
    k = oracle.gaussian_kernel_q8(7, 2.0)
    np.testing.assert_array_equal(k, [18, 34, 49, 55, 49, 34, 18])             # sums to 257, as OpenCV's does
    src = make_image(9, 90, 61, n_shapes=50)
    got = oracle.gaussian_blur(src)
    p = np.pad(src.astype(np.int64), 3, mode="reflect")                        # numpy 'reflect' == BORDER_REFLECT_101
    rows = sum(int(k[i]) * p[:, i:i + 90] for i in range(7))
    full = sum(int(k[j]) * rows[j:j + 61, :] for j in range(7))
    want = np.clip((full + 32768) >> 16, 0, 255)
    np.testing.assert_array_equal(got, want)
    assert (oracle.gaussian_blur(np.full((20, 20), 255, np.uint8)) == 255).all()   # saturation (257^2 gain)


def test_fast_atan2_accuracy_and_quadrants(oracle):
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.integers(-100000, 100000, 2)
        if x == 0 and y == 0:
            continue
        want = np.degrees(np.arctan2(float(y), float(x))) % 360.0
        got = oracle.fast_atan2(y, x)
        d = abs(got - want)
        assert min(d, 360 - d) < 0.02
    assert oracle.fast_atan2(0, 0) == 0.0
    assert oracle.fast_atan2(0, 5) == 0.0
    assert abs(oracle.fast_atan2(5, 0) - 90) < 1e-4
    assert abs(oracle.fast_atan2(0, -5) - 180) < 1e-4
    assert abs(oracle.fast_atan2(-5, 0) - 270) < 1e-4


def test_cv_round_half_even(oracle):
    L = oracle.lib()
    assert [L.ora_cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def _cv_round(x):
    """cvRound: round half to even (SSE2 cvtsd2si)."""
    return np.rint(x).astype(np.int64)


def _resize_linear_8u_integer(src, dw, dh):
    """cv::resize(INTER_LINEAR) on CV_8UC1 as OpenCV 2.4 computes it, written out independently in numpy integers: float source
    coordinate (dx + 0.5) * scale - 0.5 (double product, cast to float), cvFloor, clamped end taps, 11-bit coefficients
    saturate_cast<short>(c * 2048) (cvRound of the float product), horizontal pass S[sx] * a0 + S[sx + 1] * a1 in int, vertical pass
    (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2 (VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>)."""
    sh, sw = src.shape

    def taps(dn, sn):
        scale = np.float64(sn) / np.float64(dn)
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        f = (f - s0.astype(np.float32)).astype(np.float32)
        lo = s0 < 0
        f[lo] = 0; s0[lo] = 0
        hi = s0 >= sn - 1
        f[hi] = 0; s0[hi] = sn - 1
        a1 = _cv_round((f * np.float32(2048)).astype(np.float64))
        a0 = _cv_round(((np.float32(1) - f) * np.float32(2048)).astype(np.float64))
        return s0, np.minimum(s0 + 1, sn - 1), a0, a1
    x0, x1, a0, a1 = taps(dw, sw)
    y0, y1, b0, b1 = taps(dh, sh)
    s = src.astype(np.int64)
    Hh = s[:, x0] * a0[None, :] + s[:, x1] * a1[None, :]                       # [sh, dw] int
    out = (((b0[:, None] * (Hh[y0] >> 4)) >> 16) + ((b1[:, None] * (Hh[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("sw,sh,dw,dh", [(752, 480, 627, 400), (627, 400, 522, 333), (252, 161, 210, 134), (1241, 376, 1034, 313), (97, 61, 81, 51)])
def test_resize_equals_the_integer_restatement_bit_for_bit(oracle, sw, sh, dw, dh):
    """Pins the >>4 / >>16 / +2 >>2 rounding sequence and the 11-bit coefficient tables, not just closeness to float bilinear."""
    src = make_image(31 + sw, sw, sh, n_shapes=80)
    np.testing.assert_array_equal(oracle.resize_linear(src, dw, dh), _resize_linear_8u_integer(src, dw, dh))

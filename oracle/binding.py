"""ctypes binding of oracle/liborb_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (viorb_amd/) never does. The library is built by `make -C oracle` (also done by
__graft_entry__.build()); if the .so is missing it is built on first use (gcc is in the image).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"),
                     ("response", "f4"), ("octave", "i4"), ("class_id", "i4")])
assert KP_DTYPE.itemsize == 28


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-B" if force else "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
        L.ora_extractor_create.restype = vp
        L.ora_extractor_create.argtypes = [i, f, i, i, i]
        L.ora_extractor_destroy.argtypes = [vp]
        L.ora_extract.argtypes = [vp, vp, i, i, i, vp, vp, i]
        L.ora_extractor_tables.argtypes = [vp] * 7
        L.ora_level_size.argtypes = [vp, i, vp, vp]
        L.ora_level_copy.argtypes = [vp, i, i, vp]
        L.ora_level_keypoints.argtypes = [vp, i, i, vp, i]
        L.ora_resize_linear.argtypes = [vp, i, i, vp, i, i]
        L.ora_gaussian_blur.argtypes = [vp, i, i, vp]
        L.ora_gaussian_kernel_q8.argtypes = [i, d, vp]
        L.ora_fast_atan2.restype = f
        L.ora_fast_atan2.argtypes = [f, f]
        L.ora_cv_round.argtypes = [d]
        L.ora_fast.argtypes = [vp, i, i, i, i, i, i, i, vp, i]
        L.ora_ic_angle.restype = f
        L.ora_ic_angle.argtypes = [vp, i, i, f, f]
        L.ora_orb_descriptor.argtypes = [vp, i, i, f, f, f, vp]
        L.ora_distribute_octree.argtypes = [vp, i, i, i, i, i, i, vp, i]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Extractor:
    """Mirror of ORB_SLAM2::ORBextractor on the CPU oracle (reference include/ORBextractor.h:45-111)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = C.c_void_p(self.L.ora_extractor_create(nfeatures, scale_factor, nlevels, ini_th, min_th))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ora_extractor_destroy(self.h)
            self.h = None

    def __call__(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        hgt, w = img.shape
        cap = self.nfeatures * 2 + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.ora_extract(self.h, _p(img), w, hgt, w, _p(kps), _p(desc), cap)
        assert n <= cap
        return kps[:n].copy(), desc[:n].copy()

    def tables(self):
        n = self.nlevels
        sf, isf, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        quota = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.ora_extractor_tables(self.h, _p(sf), _p(isf), _p(s2), _p(is2), _p(quota), _p(umax))
        return dict(scale=sf, inv_scale=isf, sigma2=s2, inv_sigma2=is2, quota=quota, umax=umax)

    def level(self, l, blurred=False):
        w, h = C.c_int(), C.c_int()
        assert self.L.ora_level_size(self.h, l, C.byref(w), C.byref(h)) == 0
        out = np.zeros((h.value, w.value), np.uint8)
        if self.L.ora_level_copy(self.h, l, 1 if blurred else 0, _p(out)) == 0:
            return None                      # level had no keypoints -> reference never blurs it
        return out

    def level_keypoints(self, l, candidates=False):
        buf = np.zeros(self.nfeatures * 40 + 1024, KP_DTYPE)
        n = self.L.ora_level_keypoints(self.h, l, 0 if candidates else 1, _p(buf), len(buf))
        assert 0 <= n <= len(buf)
        return buf[:n].copy()


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().ora_resize_linear(_p(src), src.shape[1], src.shape[0], _p(dst), dw, dh)
    return dst


def gaussian_blur(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().ora_gaussian_blur(_p(src), src.shape[1], src.shape[0], _p(dst))
    return dst


def gaussian_kernel_q8(n=7, sigma=2.0):
    k = np.zeros(n, np.int32)
    lib().ora_gaussian_kernel_q8(n, sigma, _p(k))
    return k


def fast_atan2(y, x):
    return lib().ora_fast_atan2(float(y), float(x))


def fast(img, x0, y0, x1, y1, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((img.size // 4 + 16, 3), np.int32)
    n = lib().ora_fast(_p(img), img.shape[1], img.shape[0], x0, y0, x1, y1, threshold, _p(out), len(out))
    return out[:n].copy()


def ic_angle(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().ora_ic_angle(_p(img), img.shape[1], img.shape[0], float(x), float(y))


def orb_descriptor(blurred, x, y, angle):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().ora_orb_descriptor(_p(blurred), blurred.shape[1], blurred.shape[0], float(x), float(y), float(angle), _p(d))
    return d


def distribute_octree(keys, minX, maxX, minY, maxY, N):
    keys = np.ascontiguousarray(keys, KP_DTYPE)
    out = np.zeros(len(keys) + 8, KP_DTYPE)
    n = lib().ora_distribute_octree(_p(keys), len(keys), minX, maxX, minY, maxY, N, _p(out), len(out))
    return out[:n].copy()

"""Python mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-111) over the C ABI.

`ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)` and `__call__(image)` keep the
reference's argument meaning; `extract_batch_device` is the device-resident batched form used by
bench.py (torch tensors only carry device memory and the stream)."""
import ctypes as C
import numpy as np
from . import capi
from .capi import lib, check, ptr, KP_DTYPE


class ORBextractor:
    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7,
                 max_batch=1, device=0):
        self.L = lib()
        self.params = capi.ExtractorParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
        self.nlevels = nlevels
        self.max_batch = max_batch
        h = C.c_void_p()
        check(self.L.viorb_extractor_create(C.byref(self.params), max_batch, device, C.byref(h)))
        self.h = h
        cap = C.c_int()
        check(self.L.viorb_extractor_max_keypoints(self.h, C.byref(cap)))
        self.cap = cap.value

    def __del__(self):
        if getattr(self, "h", None):
            self.L.viorb_extractor_destroy(self.h)
            self.h = None

    # ---- getters (GetLevels / GetScaleFactors / ...) ------------------------------------------
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.params.scale_factor

    def tables(self):
        n = self.nlevels
        sf, isf, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        q = np.zeros(n, np.int32)
        check(self.L.viorb_extractor_tables(self.h, ptr(sf), ptr(isf), ptr(s2), ptr(is2), ptr(q)))
        return dict(scale=sf, inv_scale=isf, sigma2=s2, inv_sigma2=is2, quota=q)

    def GetScaleFactors(self):
        return self.tables()["scale"]

    def GetInverseScaleFactors(self):
        return self.tables()["inv_scale"]

    def GetScaleSigmaSquares(self):
        return self.tables()["sigma2"]

    def GetInverseScaleSigmaSquares(self):
        return self.tables()["inv_sigma2"]

    # ---- operator() ----------------------------------------------------------------------------
    def __call__(self, image, mask=None):
        """image: uint8 [h, w] numpy array (host). Returns (keypoints KP_DTYPE[n], descriptors u8[n,32])."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2          # reference asserts CV_8UC1
        image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.capacity_for(w, h)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        check(self.L.viorb_extract(self.h, ptr(image), w, h, image.strides[0], ptr(kps), ptr(desc), cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def capacity_for(self, width, height):
        """Exact keypoint bound for an image size (== self.cap unless a level has more quadtree roots than a quarter of its quota); also the
        pitch of the handle's device results for images of that size. self.cap follows the last size asked for."""
        cap = C.c_int()
        check(self.L.viorb_extractor_max_keypoints_for(self.h, int(width), int(height), C.byref(cap)))
        self.cap = cap.value
        return cap.value

    # ---- batched, device-resident ---------------------------------------------------------------
    def extract_batch_device(self, images, stream=None):
        """images: torch uint8 CUDA tensor [B, h, w] (contiguous rows). Enqueues on `stream`
        (torch.cuda.Stream or None = current stream); results stay on the device."""
        import torch
        assert images.is_cuda and images.dtype == torch.uint8 and images.dim() == 3
        B, h, w = images.shape
        assert images.stride(2) == 1
        st = stream if stream is not None else torch.cuda.current_stream(images.device)
        check(self.L.viorb_extract_batch_device(self.h, ptr(images), B, w, h, images.stride(1), images.stride(0),
                                                C.c_void_p(st.cuda_stream)))
        self._last_batch = B
        self.capacity_for(w, h)                                  # self.cap = the pitch of the device results for this size

    def download(self, b):
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int()
        check(self.L.viorb_extractor_download(self.h, b, ptr(kps), ptr(desc), self.cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def results_device(self):
        """(kps_ptr, desc_ptr, count_ptr, status_ptr, cap) raw device addresses of the last batch."""
        a, b, c, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap = C.c_int()
        check(self.L.viorb_extractor_results_device(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(cap)))
        return a.value, b.value, c.value, d.value, cap.value

    def level(self, l, b=0, blurred=False):
        """mvImagePyramid[l] of image b (host copy, un-padded)."""
        w, h = C.c_int(), C.c_int()
        check(self.L.viorb_extractor_level_download(self.h, b, l, int(blurred), None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        check(self.L.viorb_extractor_level_download(self.h, b, l, int(blurred), ptr(out), C.byref(w), C.byref(h)))
        return out

    def debug_level_points(self, l, b=0, kept=False):
        buf = np.zeros((131072, 3), np.int32)
        n = C.c_int()
        check(self.L.viorb_extractor_debug_level_points(self.h, b, l, int(kept), ptr(buf), len(buf), C.byref(n)))
        return buf[:min(n.value, len(buf))].copy()


def ComputeStereoMatches(ex_left, ex_right, bf, fx):
    """Frame::ComputeStereoMatches on image 0 of two extractors (host buffers). Returns (mvuRight, mvDepth, nmatched)."""
    u, d = np.zeros(ex_left.cap, np.float32), np.zeros(ex_left.cap, np.float32)
    n = C.c_int()
    check(lib().viorb_stereo_match(ex_left.h, ex_right.h, float(bf), float(fx), ptr(u), ptr(d), ex_left.cap, C.byref(n)))
    return u, d, n.value


def octree_host(keys_xyr, width, height, N):
    """Host test hook: the product's flat-array DistributeOctTree on (x, y, response) int triples."""
    k = np.asarray(keys_xyr, np.int64).reshape(-1, 3)
    packed = (k[:, 0] | (k[:, 1] << 12) | (k[:, 2] << 24)).astype(np.uint32)
    out = np.zeros(len(packed) + 16, np.uint32)
    n = C.c_int()
    check(lib().viorb_debug_octree_host(ptr(packed), len(packed), width, height, N, ptr(out), len(out), C.byref(n)))
    o = out[:n.value].astype(np.int64)
    return np.stack([o & 0xfff, (o >> 12) & 0xfff, o >> 24], axis=1).astype(np.int32)

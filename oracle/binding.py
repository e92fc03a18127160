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
        if n > len(buf):                     # an image of noise: more candidates than the usual buffer
            buf = np.zeros(n, KP_DTYPE)
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


# ==================================================================================================
# Visual-inertial part (oracle/vio.{h,cpp}); flat float64 layouts documented in oracle_capi.cpp
# ==================================================================================================
NS_LEN, PREINT_LEN, CAM_LEN = 22, 142, 16
_vio_ready = False


def _vio():
    global _vio_ready
    L = lib()
    if not _vio_ready:
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        L.ora_preintegrate.argtypes = [vp, i, vp, vp, d, d, vp]
        L.ora_preint_update.argtypes = [vp, vp, vp, d]
        L.ora_update_ns.argtypes = [vp, vp, vp]
        L.ora_ns_inc_pvr.argtypes = [vp, vp]
        L.ora_predict_navstate.argtypes = [vp, vp, vp, vp]
        L.ora_pose_from_navstate.argtypes = [vp, vp, vp]
        L.ora_so3_exp.argtypes = [vp, vp]
        L.ora_so3_log.argtypes = [vp, vp]
        L.ora_so3_matrix.argtypes = [vp, vp]
        L.ora_so3_from_matrix.argtypes = [vp, vp]
        L.ora_jacobian_r.argtypes = [vp, vp, i]
        L.ora_edge_pvr.argtypes = [vp] * 9
        L.ora_edge_proj.argtypes = [vp] * 5
        L.ora_edge_prior.argtypes = [vp] * 6
        L.ora_pose_opt_se3.argtypes = [vp, vp, vp, i, vp, vp, vp]
        L.ora_pose_opt_vi_kf.argtypes = [vp, vp, vp, vp, vp, vp, i, i, vp, vp, vp, vp, vp, i]
        L.ora_pose_opt_vi_frame.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, vp, i, i, vp, vp, vp, vp, vp, vp, vp, i]
        _vio_ready = True
    return L


def _f64(a, n=None):
    a = np.ascontiguousarray(a, np.float64)
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def preintegrate(samples, bg, ba, t_last, t_cur):
    """samples: [n,7] = gyro3, acc3, t. Returns preint[142]."""
    s = _f64(samples).reshape(-1, 7)
    out = np.zeros(PREINT_LEN)
    _vio().ora_preintegrate(_p(s), len(s), _p(_f64(bg, 3)), _p(_f64(ba, 3)), float(t_last), float(t_cur), _p(out))
    return out


def preint_update(preint, omega, acc, dt):
    p = _f64(preint, PREINT_LEN).copy()
    _vio().ora_preint_update(_p(p), _p(_f64(omega, 3)), _p(_f64(acc, 3)), float(dt))
    return p


def update_ns(ns, preint, gw):
    n = _f64(ns, NS_LEN).copy()
    _vio().ora_update_ns(_p(n), _p(_f64(preint, PREINT_LEN)), _p(_f64(gw, 3)))
    return n


def predict_navstate(last_ns, preint, gw):
    """PredictNavStateByIMU: SetInitialNavStateAndBias(last) + UpdateNavState."""
    out = np.zeros(NS_LEN)
    _vio().ora_predict_navstate(_p(_f64(last_ns, NS_LEN)), _p(_f64(preint, PREINT_LEN)), _p(_f64(gw, 3)), _p(out))
    return out


def pose_from_navstate(ns, cam):
    """Frame::UpdatePoseFromNS -> float pose12 = Rcw(9) tcw(3)."""
    out = np.zeros(12, np.float32)
    _vio().ora_pose_from_navstate(_p(_f64(ns, NS_LEN)), _p(_f64(cam, CAM_LEN)), _p(out))
    return out


def ns_inc_pvr(ns, u9):
    n = _f64(ns, NS_LEN).copy()
    _vio().ora_ns_inc_pvr(_p(n), _p(_f64(u9, 9)))
    return n


def so3_exp(w):
    q = np.zeros(4); _vio().ora_so3_exp(_p(_f64(w, 3)), _p(q)); return q


def so3_log(q):
    w = np.zeros(3); _vio().ora_so3_log(_p(_f64(q, 4)), _p(w)); return w


def so3_matrix(q):
    R = np.zeros(9); _vio().ora_so3_matrix(_p(_f64(q, 4)), _p(R)); return R.reshape(3, 3)


def so3_from_matrix(R):
    q = np.zeros(4); _vio().ora_so3_from_matrix(_p(_f64(R, 9)), _p(q)); return q


def jacobian_r(w, inverse=False):
    J = np.zeros(9); _vio().ora_jacobian_r(_p(_f64(w, 3)), _p(J), int(inverse)); return J.reshape(3, 3)


def edge_pvr(ni, nj, nb, preint, gw, jac=True):
    e, Ji, Jj, Jb = np.zeros(9), np.zeros(81), np.zeros(81), np.zeros(27)
    _vio().ora_edge_pvr(_p(_f64(ni, 22)), _p(_f64(nj, 22)), _p(_f64(nb, 22)), _p(_f64(preint, 142)), _p(_f64(gw, 3)),
                        _p(e), _p(Ji) if jac else None, _p(Jj), _p(Jb))
    return e, Ji.reshape(9, 9), Jj.reshape(9, 9), Jb.reshape(9, 3)


def edge_proj(ns, cam, obs, jac=True):
    e, J = np.zeros(2), np.zeros(18)
    _vio().ora_edge_proj(_p(_f64(ns, 22)), _p(_f64(cam, 16)), _p(_f64(obs, 6)), _p(e), _p(J) if jac else None)
    return e, J.reshape(2, 9)


def edge_prior(pvr, bias, prior, jac=True):
    e, Jp, Jb = np.zeros(12), np.zeros(108), np.zeros(36)
    _vio().ora_edge_prior(_p(_f64(pvr, 22)), _p(_f64(bias, 22)), _p(_f64(prior, 22)), _p(e), _p(Jp) if jac else None, _p(Jb))
    return e, Jp.reshape(12, 9), Jb.reshape(12, 3)


def _result(ns, ns_last, oc, ol, marg, info, trace):
    d = (C.c_int * 2)(); _vio().ora_pose_opt_diagnostics(d)
    return dict(rejected_rounds=int(d[0]), stale_verdicts=int(d[1]), ns=ns, ns_last=ns_last, outlier_cur=oc, outlier_last=ol, marg_cov_inv=marg.reshape(12, 12),
                n_inliers=int(info[0]), final_chi2=float(info[1]), lm_iterations=int(info[2]),
                chi2_trace=trace[:int(info[3])].copy())


def pose_opt_vi_kf(cur, kf, preint, gw, cam, obs, marg=False):
    """Optimizer::PoseOptimization(Frame*, KeyFrame*, ...) on the oracle. obs: [n,6] = Pw3,u,v,invSigma2."""
    obs = _f64(obs).reshape(-1, 6)
    ns, oc, mg, info, tr = np.zeros(22), np.zeros(max(len(obs), 1), np.uint8), np.zeros(144), np.zeros(4), np.zeros(64)
    _vio().ora_pose_opt_vi_kf(_p(_f64(cur, 22)), _p(_f64(kf, 22)), _p(_f64(preint, 142)), _p(_f64(gw, 3)), _p(_f64(cam, 16)),
                              _p(obs), len(obs), int(marg), _p(ns), _p(oc), _p(mg), _p(info), _p(tr), len(tr))
    return _result(ns, None, oc[:len(obs)], None, mg, info, tr)


def pose_opt_vi_frame(cur, last, prior, marg_cov_inv, preint, gw, cam, obs_cur, obs_last, marg=False):
    """Optimizer::PoseOptimization(Frame*, Frame*, ...) on the oracle."""
    oc_, ol_ = _f64(obs_cur).reshape(-1, 6), _f64(obs_last).reshape(-1, 6)
    ns, nl = np.zeros(22), np.zeros(22)
    oc, ol = np.zeros(max(len(oc_), 1), np.uint8), np.zeros(max(len(ol_), 1), np.uint8)
    mg, info, tr = np.zeros(144), np.zeros(4), np.zeros(64)
    _vio().ora_pose_opt_vi_frame(_p(_f64(cur, 22)), _p(_f64(last, 22)), _p(_f64(prior, 22)), _p(_f64(marg_cov_inv, 144)),
                                 _p(_f64(preint, 142)), _p(_f64(gw, 3)), _p(_f64(cam, 16)), _p(oc_), len(oc_), _p(ol_), len(ol_),
                                 int(marg), _p(ns), _p(nl), _p(oc), _p(ol), _p(mg), _p(info), _p(tr), len(tr))
    return _result(ns, nl, oc[:len(oc_)], ol[:len(ol_)], mg, info, tr)


# ==================================================================================================
# Matcher (oracle/orb_matcher.{h,cpp})
# ==================================================================================================
_match_ready = False


def _match():
    global _match_ready
    L = lib()
    if not _match_ready:
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        L.ora_descriptor_distance.argtypes = [vp, vp]
        L.ora_match_bruteforce.argtypes = [vp, i, vp, i, vp, vp, vp]
        L.ora_match_bruteforce.restype = None
        L.ora_frame_grid.argtypes = [vp, i, f, f, f, f, vp, vp]
        L.ora_features_in_area.argtypes = [vp, i, f, f, f, f, f, f, f, i, i, vp, i]
        L.ora_search_by_projection_frame.argtypes = [vp, vp, i, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, f, i, vp]
        L.ora_search_by_projection_frame_stereo.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, f, f, vp, i, vp, vp, vp, vp, vp, f, i, vp]
        L.ora_search_local_points.argtypes = [vp, vp, i, vp, vp, vp, vp, i, f, i, vp, vp, vp, f, f, vp, vp, vp]
        L.ora_search_local_points_stereo.argtypes = [vp, vp, vp, f, i, vp, vp, vp, vp, i, f, i, vp, vp, vp, f, f, vp, vp, vp, vp]
        _match_ready = True
    return L


def descriptor_distance(a, b):
    return _match().ora_descriptor_distance(_p(np.ascontiguousarray(a, np.uint8)), _p(np.ascontiguousarray(b, np.uint8)))


def match_bruteforce(q, c):
    """(best, second, idx) per query over all candidates (strict '<': first candidate wins a tie)."""
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); c = np.ascontiguousarray(c, np.uint8).reshape(-1, 32)
    best, second, idx = (np.zeros(len(q), np.int32) for _ in range(3))
    _match().ora_match_bruteforce(_p(q), len(q), _p(c), len(c), _p(best), _p(second), _p(idx))
    return best, second, idx


def frame_grid(kps, bounds):
    """CSR of Frame::mGrid in storage order [ix][iy]: (cell_start[64*48+1], cell_idx[nbinned])."""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    cs = np.zeros(64 * 48 + 1, np.int32)
    ci = np.zeros(max(len(kps), 1), np.int32)
    n = _match().ora_frame_grid(_p(kps), len(kps), *[float(b) for b in bounds], _p(cs), _p(ci))
    return cs, ci[:n].copy()


def features_in_area(kps, bounds, x, y, r, min_level=-1, max_level=-1):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    out = np.zeros(max(len(kps), 1), np.int32)
    n = _match().ora_features_in_area(_p(kps), len(kps), *[float(b) for b in bounds], float(x), float(y), float(r),
                                      int(min_level), int(max_level), _p(out), len(out))
    return out[:n].copy()


def search_by_projection_frame(cur_kps, cur_desc, bounds, pose12, intr4, scale_factors, last_flags, last_Pw, last_mp_desc,
                               last_octave, last_angle, th, check_ori=True, cur_match=None):
    """ORBmatcher::SearchByProjection(Cur, Last, th, mono). Returns (nmatches, cur_match[Ncur])."""
    cur_kps = np.ascontiguousarray(cur_kps, KP_DTYPE)
    n = len(cur_kps)
    m = np.full(max(n, 1), -1, np.int32) if cur_match is None else np.ascontiguousarray(cur_match, np.int32).copy()
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    nm = _match().ora_search_by_projection_frame(
        _p(cur_kps), _p(np.ascontiguousarray(cur_desc, np.uint8)), n, _p(f32(bounds)), _p(f32(pose12)), _p(f32(intr4)),
        _p(f32(scale_factors)), len(last_flags), _p(np.ascontiguousarray(last_flags, np.uint8)), _p(f32(last_Pw)),
        _p(np.ascontiguousarray(last_mp_desc, np.uint8)), _p(np.ascontiguousarray(last_octave, np.int32)), _p(f32(last_angle)),
        float(th), int(check_ori), _p(m))
    return nm, m[:n]


def search_by_projection_frame_stereo(cur_kps, cur_desc, cur_uright, bounds, pose12, last_pose12, intr4, bf, mb, scale_factors, last_flags, last_Pw,
                                      last_mp_desc, last_octave, last_angle, th, check_ori=True):
    """ORBmatcher::SearchByProjection(Cur, Last, th, bMono=false): stereo / RGB-D branch. Returns (nmatches, cur_match[Ncur])."""
    cur_kps = np.ascontiguousarray(cur_kps, KP_DTYPE)
    n = len(cur_kps)
    m = np.full(max(n, 1), -1, np.int32)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    nm = _match().ora_search_by_projection_frame_stereo(
        _p(cur_kps), _p(np.ascontiguousarray(cur_desc, np.uint8)), _p(f32(cur_uright)), n, _p(f32(bounds)), _p(f32(pose12)), _p(f32(last_pose12)),
        _p(f32(intr4)), float(bf), float(mb), _p(f32(scale_factors)), len(last_flags), _p(np.ascontiguousarray(last_flags, np.uint8)), _p(f32(last_Pw)),
        _p(np.ascontiguousarray(last_mp_desc, np.uint8)), _p(np.ascontiguousarray(last_octave, np.int32)), _p(f32(last_angle)),
        float(th), int(check_ori), _p(m))
    return nm, m[:n]


def search_local_points(cur_kps, cur_desc, bounds, pose12, intr4, scale_factors, log_scale_factor, pts_f, pts_flags, pts_desc,
                        th, nnratio, cur_owner_obs, cur_uright=None, bf=0.0):
    """Tracking::SearchLocalPoints (isInFrustum per point) + ORBmatcher::SearchByProjection(F, vpMapPoints, th).
    pts_f [n,8] = Pw3 normal3 minDist maxDist. Returns (nmatches, match[Ncur], frustum[n,5]); with cur_uright (mvuRight of a
    stereo / RGB-D frame) and bf (mbf) the right-coordinate gate of ORBmatcher.cc:91-97 applies and a fourth value, mTrackProjXR [n], is returned."""
    cur_kps = np.ascontiguousarray(cur_kps, KP_DTYPE)
    n = len(cur_kps)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    pts_f = f32(pts_f).reshape(-1, 8)
    m = np.full(max(n, 1), -1, np.int32)
    fr = np.zeros((max(len(pts_f), 1), 5), np.float32)
    sf = f32(scale_factors)
    if cur_uright is not None:
        xr = np.zeros(max(len(pts_f), 1), np.float32)
        nm = _match().ora_search_local_points_stereo(_p(cur_kps), _p(np.ascontiguousarray(cur_desc, np.uint8)), _p(f32(cur_uright)), float(bf), n, _p(f32(bounds)),
                                                     _p(f32(pose12)), _p(f32(intr4)), _p(sf), len(sf), float(log_scale_factor), len(pts_f), _p(pts_f),
                                                     _p(np.ascontiguousarray(pts_flags, np.uint8)), _p(np.ascontiguousarray(pts_desc, np.uint8)),
                                                     float(th), float(nnratio), _p(np.ascontiguousarray(cur_owner_obs, np.uint8)), _p(m), _p(fr), _p(xr))
        return nm, m[:n], fr[:len(pts_f)], xr[:len(pts_f)]
    nm = _match().ora_search_local_points(_p(cur_kps), _p(np.ascontiguousarray(cur_desc, np.uint8)), n, _p(f32(bounds)), _p(f32(pose12)),
                                          _p(f32(intr4)), _p(sf), len(sf), float(log_scale_factor), len(pts_f), _p(pts_f),
                                          _p(np.ascontiguousarray(pts_flags, np.uint8)), _p(np.ascontiguousarray(pts_desc, np.uint8)),
                                          float(th), float(nnratio), _p(np.ascontiguousarray(cur_owner_obs, np.uint8)), _p(m), _p(fr))
    return nm, m[:n], fr[:len(pts_f)]


def pose_opt_se3(pose12, intr5, obs7):
    """Optimizer::PoseOptimization(Frame*) (vision only). obs7 [n,7] = Xw3 u v ur invSigma2 (ur < 0 = mono edge),
    intr5 = fx fy cx cy bf. Returns dict(pose12 float32, outlier, n_inliers, final_chi2, lm_iterations)."""
    obs7 = _f64(obs7).reshape(-1, 7)
    out, fl, info = np.zeros(12, np.float32), np.zeros(max(len(obs7), 1), np.uint8), np.zeros(3)
    _vio().ora_pose_opt_se3(_p(np.ascontiguousarray(pose12, np.float32)), _p(_f64(intr5, 5)), _p(obs7), len(obs7), _p(out), _p(fl), _p(info))
    d = (C.c_int * 2)(); _vio().ora_pose_opt_diagnostics(d)
    return dict(pose12=out, outlier=fl[:len(obs7)], n_inliers=int(info[0]), final_chi2=float(info[1]), lm_iterations=int(info[2]),
                rejected_rounds=int(d[0]), stale_verdicts=int(d[1]))


def stereo_match(ex_left, ex_right, kl, dl, kr, dr, bf, fx):
    """Frame::ComputeStereoMatches on the pyramids the two oracle Extractor objects hold after their last call.
    Returns (uRight[N], depth[N], best_sad[N]) with -1 where unmatched."""
    L = lib()
    L.ora_stereo_match.argtypes = [C.c_void_p] * 3 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    kl = np.ascontiguousarray(kl, KP_DTYPE); kr = np.ascontiguousarray(kr, KP_DTYPE)
    n = len(kl)
    u, d, sad = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.int32)
    L.ora_stereo_match(ex_left.h, ex_right.h, _p(kl), _p(np.ascontiguousarray(dl, np.uint8)), n, _p(kr), _p(np.ascontiguousarray(dr, np.uint8)), len(kr),
                       float(bf), float(fx), _p(u), _p(d), _p(sad))
    return u[:n], d[:n], sad[:n]


def local_ba(kfs, n_local, prev_kf, preint, points, edge_idx, edge_obs, gw, cam, stop=None):
    """Optimizer::LocalBundleAdjustmentNavState on the oracle. kfs [NK,22] local first; preint [W,142]; points [NP,3];
    edge_idx [NE,2] int32 (point, kf) grouped by point; edge_obs [NE,3] = u v invSigma2."""
    L = lib()
    L.ora_local_ba.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7
    kfs = _f64(kfs).reshape(-1, 22); preint = _f64(preint).reshape(-1, 142); points = _f64(points).reshape(-1, 3)
    ei = np.ascontiguousarray(edge_idx, np.int32).reshape(-1, 2); eo = _f64(edge_obs).reshape(-1, 3)
    ko, po = np.zeros((n_local, 22)), np.zeros_like(points)
    er, info = np.zeros(max(len(ei), 1), np.uint8), np.zeros(6)
    st = np.ascontiguousarray(stop, np.int32) if stop is not None else None
    L.ora_local_ba(_p(kfs), len(kfs), n_local, prev_kf, _p(preint), _p(points), len(points), _p(ei), _p(eo), len(ei), _p(_f64(gw, 3)),
                   _p(_f64(cam, 16)), _p(st) if st is not None else None, _p(ko), _p(po), _p(er), _p(info))
    return dict(kfs=ko, points=po, erase=er[:len(ei)], chi2_first=info[0], chi2_final=info[1], its_first=int(info[2]), its_second=int(info[3]))


def bow_transform(voc, desc, levelsup=4):
    """DBoW2 TemplatedVocabulary::transform (TF_IDF, L1) over a flat vocabulary dict (viorb_amd.synth.make_vocabulary).
    Returns dict(word, weight, node per feature; bow_ids, bow_vals = the L1-normalised BowVector)."""
    L = lib()
    L.ora_bow_transform.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int] + [C.c_void_p] * 5
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); n = len(desc)
    word, weight, node = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1)), np.zeros(max(n, 1), np.int32)
    bi, bv = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1))
    k = L.ora_bow_transform(len(voc["word_id"]), int(voc["L"]), _p(voc["child_start"]), _p(voc["child_ids"]), _p(voc["desc"]), _p(voc["word_id"]),
                            _p(voc["weight"]), _p(desc), n, levelsup, _p(word), _p(weight), _p(node), _p(bi), _p(bv))
    return dict(word=word[:n], weight=weight[:n], node=node[:n], bow_ids=bi[:k], bow_vals=bv[:k])


def search_by_bow(kf_desc, kf_angle, kf_node, kf_has_point, f_desc, f_angle, f_node, nnratio=0.7, check_orientation=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...). Returns (nmatches, match[nF]) with match = key-frame feature index or -1."""
    L = lib()
    L.ora_search_by_bow.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
    kd = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32); fd = np.ascontiguousarray(f_desc, np.uint8).reshape(-1, 32)
    m = np.full(max(len(fd), 1), -1, np.int32)
    n = L.ora_search_by_bow(_p(kd), _p(np.ascontiguousarray(kf_angle, np.float32)), _p(np.ascontiguousarray(kf_node, np.int32)),
                            _p(np.ascontiguousarray(kf_has_point, np.uint8)), len(kd), _p(fd), _p(np.ascontiguousarray(f_angle, np.float32)),
                            _p(np.ascontiguousarray(f_node, np.int32)), len(fd), float(nnratio), int(check_orientation), _p(m))
    return n, m[:len(fd)]


def search_for_triangulation(k1, d1, hp1, ur1, node1, k2, d2, hp2, ur2, node2, F12, Cw1, pose12_2, intr4, sf2, level_sigma2_2,
                             only_stereo=False, check_orientation=True):
    """ORBmatcher::SearchForTriangulation. Returns (nmatches, match12[N1]) with match12[i1] = i2 or -1."""
    L = lib()
    L.ora_search_for_triangulation.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]
    k1 = np.ascontiguousarray(k1, KP_DTYPE); k2 = np.ascontiguousarray(k2, KP_DTYPE)
    u8 = lambda a: np.ascontiguousarray(a, np.uint8); f32 = lambda a: np.ascontiguousarray(a, np.float32); i32 = lambda a: np.ascontiguousarray(a, np.int32)
    m = np.full(max(len(k1), 1), -1, np.int32)
    n = L.ora_search_for_triangulation(_p(k1), _p(u8(d1)), _p(u8(hp1)), _p(f32(ur1)), _p(i32(node1)), len(k1), _p(k2), _p(u8(d2)), _p(u8(hp2)),
                                       _p(f32(ur2)), _p(i32(node2)), len(k2), _p(f32(F12)), _p(f32(Cw1)), _p(f32(pose12_2)), _p(f32(intr4)),
                                       _p(f32(sf2)), _p(f32(level_sigma2_2)), int(only_stereo), int(check_orientation), _p(m))
    return n, m[:len(k1)]


def fuse(kps, desc, uright, bounds, pose12, intr5, scale_factors, inv_level_sigma2, log_scale_factor, pts_f, pts_valid, pts_desc, th=3.0):
    """ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th): best_idx per map point (-1: not fused). Returns (nFused, best_idx)."""
    L = lib()
    L.ora_fuse.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_float, C.c_int] + [C.c_void_p] * 3 + [C.c_float, C.c_void_p]
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    pts_f = f32(pts_f).reshape(-1, 8); sf = f32(scale_factors)
    bi = np.full(max(len(pts_f), 1), -1, np.int32)
    n = L.ora_fuse(_p(kps), _p(np.ascontiguousarray(desc, np.uint8)), _p(f32(uright)), len(kps), _p(f32(bounds)), _p(f32(pose12)), _p(f32(intr5)),
                   _p(sf), _p(f32(inv_level_sigma2)), len(sf), float(log_scale_factor), len(pts_f), _p(pts_f), _p(np.ascontiguousarray(pts_valid, np.uint8)),
                   _p(np.ascontiguousarray(pts_desc, np.uint8)), float(th), _p(bi))
    return n, bi[:len(pts_f)]


def local_ba_se3(kfs, n_local, points, edge_idx, edge_obs, intr5, stop=None):
    """Vision-only Optimizer::LocalBundleAdjustment. kfs [NK,7] = qx qy qz qw tx ty tz (Tcw), free ones first; points [NP,3];
    edge_idx [NE,2] int32 (point, kf) grouped by point; edge_obs [NE,4] = u v uRight(<0 mono) invSigma2; intr5 = fx fy cx cy bf."""
    L = lib()
    L.ora_local_ba_se3.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6
    kfs = _f64(kfs).reshape(-1, 7); points = _f64(points).reshape(-1, 3)
    ei = np.ascontiguousarray(edge_idx, np.int32).reshape(-1, 2); eo = _f64(edge_obs).reshape(-1, 4)
    ko, po = np.zeros((n_local, 7)), np.zeros_like(points)
    er, info = np.zeros(max(len(ei), 1), np.uint8), np.zeros(6)
    st = np.ascontiguousarray(stop, np.int32) if stop is not None else None
    L.ora_local_ba_se3(_p(kfs), len(kfs), n_local, _p(points), len(points), _p(ei), _p(eo), len(ei), _p(_f64(intr5, 5)), _p(st) if st is not None else None,
                       _p(ko), _p(po), _p(er), _p(info))
    return dict(kfs=ko, points=po, erase=er[:len(ei)], chi2_first=info[0], chi2_final=info[1], its_first=int(info[2]), its_second=int(info[3]))


def undistort_points(xy, K4, dist5):
    """cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK) of Frame::UndistortKeyPoints (reference src/Frame.cc:584-614): xy [n,2] float32
    pixels -> undistorted float32 pixels. K4 = fx fy cx cy, dist5 = k1 k2 p1 p2 k3 (float32, as Frame::mK / mDistCoef)."""
    L = lib()
    L.ora_undistort_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.empty_like(xy)
    L.ora_undistort_points(_p(xy), len(xy), _p(np.ascontiguousarray(K4, np.float32)), _p(np.ascontiguousarray(dist5, np.float32)), _p(out))
    return out


def image_bounds(width, height, K4, dist5):
    """Frame::ComputeImageBounds (reference src/Frame.cc:616-644): float32 [mnMinX, mnMaxX, mnMinY, mnMaxY]."""
    L = lib()
    L.ora_image_bounds.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    b = np.zeros(4, np.float32)
    L.ora_image_bounds(int(width), int(height), _p(np.ascontiguousarray(K4, np.float32)), _p(np.ascontiguousarray(dist5, np.float32)), _p(b))
    return b

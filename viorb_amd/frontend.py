"""Python mirror of the per-frame tracking calls over the C ABI (include/viorb.h, "front-end" section):
ORBmatcher::SearchByProjection(Frame, Frame), Frame grid, IMU pre-integration + NavState prediction and
Optimizer::PoseOptimization with NavState edges. `Frontend` is the batched, device-resident form (torch
tensors carry device memory only); the module-level functions are the host-buffer drop-ins."""
import ctypes as C
import numpy as np
from . import capi
from .capi import lib, check, ptr

GRID_CELLS = 64 * 48


def descriptor_distance(a, b):
    """ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:1648-1664)."""
    return lib().viorb_descriptor_distance(ptr(np.ascontiguousarray(a, np.uint8)), ptr(np.ascontiguousarray(b, np.uint8)))


def match_bruteforce(q_desc, c_desc):
    """Brute-force Hamming matcher (viorb_match_bruteforce): (best, second, idx) per query over ALL candidates, strict '<'
    (the reference's bestDist1 / bestDist2 / bestIdx scan, src/ORBmatcher.cc:204-222, with DescriptorDistance :1648-1664)."""
    q = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32); c = np.ascontiguousarray(c_desc, np.uint8).reshape(-1, 32)
    best, second, idx = (np.zeros(max(len(q), 1), np.int32) for _ in range(3))
    check(lib().viorb_match_bruteforce(ptr(q), len(q), ptr(c), len(c), ptr(best), ptr(second), ptr(idx)))
    return best[:len(q)], second[:len(q)], idx[:len(q)]


def UndistortKeyPoints(xy, intr4, dist_coef):
    """Frame::UndistortKeyPoints' cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK) (reference src/Frame.cc:584-614): xy [n,2]
    float32 pixels -> undistorted float32 pixels; intr4 = fx fy cx cy, dist_coef = k1 k2 p1 p2 [k3]."""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    d = np.zeros(5, np.float32); d[:len(dist_coef)] = np.asarray(dist_coef, np.float32)[:5]
    out = np.empty_like(xy)
    check(lib().viorb_undistort_points(ptr(xy), len(xy), ptr(np.ascontiguousarray(intr4, np.float32)), ptr(d), ptr(out)))
    return out


def ComputeImageBounds(width, height, intr4, dist_coef):
    """Frame::ComputeImageBounds (reference src/Frame.cc:616-644): float32 [mnMinX, mnMaxX, mnMinY, mnMaxY]."""
    d = np.zeros(5, np.float32); d[:len(dist_coef)] = np.asarray(dist_coef, np.float32)[:5]
    b = np.zeros(4, np.float32)
    check(lib().viorb_image_bounds(int(width), int(height), ptr(np.ascontiguousarray(intr4, np.float32)), ptr(d), ptr(b)))
    return b


class ORBmatcher:
    """Mirror of ORB_SLAM2::ORBmatcher for the frame-side searches (reference include/ORBmatcher.h:37-102)."""
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)

    DescriptorDistance = staticmethod(descriptor_distance)

    def SearchByProjection(self, cur_kps, cur_desc, bounds, pose12, intr4, scale_factors, last_kps, last_flags, last_Pw,
                           last_mp_desc, th, bMono=True, cur_uright=None, last_pose12=None, bf=0.0, mb=0.0):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono) on SoA host arrays. bMono=False (stereo / RGB-D) also needs
        CurrentFrame.mvuRight, LastFrame.mTcw (Rlw, tlw), mbf and mb (reference src/ORBmatcher.cc:1346-1349, 1385-1410).
        Returns (nmatches, cur_match[Ncur]) with cur_match[i2] = last-frame index or -1."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        ck = np.ascontiguousarray(cur_kps, capi.KP_DTYPE); lk = np.ascontiguousarray(last_kps, capi.KP_DTYPE)
        match = np.full(max(len(ck), 1), -1, np.int32)
        nm = C.c_int()
        sf = f32(scale_factors)
        if not bMono:
            assert cur_uright is not None and last_pose12 is not None
            check(lib().viorb_search_by_projection_frame_stereo(ptr(ck), ptr(np.ascontiguousarray(cur_desc, np.uint8)), ptr(f32(cur_uright)), len(ck),
                                                                ptr(f32(bounds)), ptr(f32(pose12)), ptr(f32(last_pose12)), ptr(f32(intr4)), float(bf),
                                                                float(mb), ptr(sf), len(sf), ptr(lk), len(lk),
                                                                ptr(np.ascontiguousarray(last_flags, np.uint8)), ptr(f32(last_Pw)),
                                                                ptr(np.ascontiguousarray(last_mp_desc, np.uint8)), float(th),
                                                                int(self.mbCheckOrientation), ptr(match), C.byref(nm)))
            return nm.value, match[:len(ck)]
        check(lib().viorb_search_by_projection_frame(ptr(ck), ptr(np.ascontiguousarray(cur_desc, np.uint8)), len(ck), ptr(f32(bounds)),
                                                     ptr(f32(pose12)), ptr(f32(intr4)), ptr(sf), len(sf), ptr(lk), len(lk),
                                                     ptr(np.ascontiguousarray(last_flags, np.uint8)), ptr(f32(last_Pw)),
                                                     ptr(np.ascontiguousarray(last_mp_desc, np.uint8)), float(th),
                                                     int(self.mbCheckOrientation), ptr(match), C.byref(nm)))
        return nm.value, match[:len(ck)]


def SearchLocalPoints(cur_kps, cur_desc, bounds, pose12, intr4, scale_factors, pts_f, pts_flags, pts_desc, th=1.0, nnratio=0.8, cur_owner_obs=None,
                      want_frustum=False, cur_uright=None, bf=0.0):
    """Tracking::SearchLocalPoints' matcher call: Frame::isInFrustum + ORBmatcher(nnratio).SearchByProjection(F, vpMapPoints, th)
    (reference src/Tracking.cc:1904-1958, src/ORBmatcher.cc:45-129), host arrays. Returns (nmatches, match[N]) (+ frustum[npts,5]).
    cur_uright (F.mvuRight) + bf (F.mbf): a stereo / RGB-D frame — the right-coordinate gate of ORBmatcher.cc:91-97 applies and the frustum
    result gains mTrackProjXR as a third return value ((nmatches, match, frustum, proj_xr) with want_frustum)."""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    ck = np.ascontiguousarray(cur_kps, capi.KP_DTYPE)
    pf = f32(pts_f).reshape(-1, 8); sf = f32(scale_factors)
    own = np.zeros(max(len(ck), 1), np.uint8) if cur_owner_obs is None else np.ascontiguousarray(cur_owner_obs, np.uint8)
    match = np.full(max(len(ck), 1), -1, np.int32); nm = C.c_int()
    fr = np.zeros((max(len(pf), 1), 5), np.float32) if want_frustum else None
    if cur_uright is not None:
        ur = f32(cur_uright); assert len(ur) >= len(ck)
        xr = np.zeros(max(len(pf), 1), np.float32) if want_frustum else None
        check(lib().viorb_search_by_projection_points_stereo(ptr(ck), ptr(np.ascontiguousarray(cur_desc, np.uint8)), ptr(ur), float(bf), len(ck), ptr(f32(bounds)),
                                                             ptr(f32(pose12)), ptr(f32(intr4)), ptr(sf), len(sf), ptr(pf), ptr(np.ascontiguousarray(pts_flags, np.uint8)),
                                                             ptr(np.ascontiguousarray(pts_desc, np.uint8)), len(pf), float(th), float(nnratio), ptr(own), ptr(match),
                                                             C.byref(nm), ptr(fr) if fr is not None else None, ptr(xr) if xr is not None else None))
        return (nm.value, match[:len(ck)], fr[:len(pf)], xr[:len(pf)]) if want_frustum else (nm.value, match[:len(ck)])
    check(lib().viorb_search_by_projection_points(ptr(ck), ptr(np.ascontiguousarray(cur_desc, np.uint8)), len(ck), ptr(f32(bounds)), ptr(f32(pose12)),
                                                  ptr(f32(intr4)), ptr(sf), len(sf), ptr(pf), ptr(np.ascontiguousarray(pts_flags, np.uint8)),
                                                  ptr(np.ascontiguousarray(pts_desc, np.uint8)), len(pf), float(th), float(nnratio), ptr(own), ptr(match),
                                                  C.byref(nm), ptr(fr) if fr is not None else None))
    return (nm.value, match[:len(ck)], fr[:len(pf)]) if want_frustum else (nm.value, match[:len(ck)])


def preintegrate(imu, bg, ba, t_last, t_cur):
    """IMU pre-integration between two frames (Frame::ComputeIMUPreIntSinceLastFrame). Returns preint[142]."""
    imu = np.ascontiguousarray(imu, np.float64).reshape(-1, 7)
    out = np.zeros(142)
    check(lib().viorb_preintegrate(ptr(imu), len(imu), ptr(np.ascontiguousarray(bg, np.float64)),
                                   ptr(np.ascontiguousarray(ba, np.float64)), float(t_last), float(t_cur), ptr(out)))
    return out


def PoseOptimization(cur_ns, last_ns, preint, gw, cam, obs_cur, obs_last=None, prior_ns=None, marg_cov_inv=None,
                     last_is_keyframe=True, bComputeMarg=False):
    """Optimizer::PoseOptimization(Frame*, KeyFrame*|Frame*, imupreint, gw, bComputeMarg)
    (reference src/Optimizer.cc:323-1112) with host buffers. Returns a dict like the oracle binding's."""
    f = lambda a: np.ascontiguousarray(a, np.float64)
    oc = f(obs_cur).reshape(-1, 6)
    ol = f(obs_last).reshape(-1, 6) if obs_last is not None else np.zeros((0, 6))
    variant = 0 if last_is_keyframe else 1
    ns, nl, mg, info = np.zeros(22), np.zeros(22), np.zeros(144), np.zeros(4)
    fc, fl = np.zeros(max(len(oc), 1), np.uint8), np.zeros(max(len(ol), 1), np.uint8)
    check(lib().viorb_pose_opt_vi(variant, int(bComputeMarg), ptr(f(cur_ns)), ptr(f(last_ns)),
                                  ptr(f(prior_ns)) if prior_ns is not None else None,
                                  ptr(f(marg_cov_inv)) if marg_cov_inv is not None else None,
                                  ptr(f(preint)), ptr(f(gw)), ptr(f(cam)), ptr(oc), len(oc), ptr(ol), len(ol),
                                  ptr(ns), ptr(nl), ptr(fc), ptr(fl), ptr(mg), ptr(info)))
    return dict(ns=ns, ns_last=nl, outlier_cur=fc[:len(oc)], outlier_last=fl[:len(ol)], marg_cov_inv=mg.reshape(12, 12),
                n_inliers=int(info[0]), final_chi2=float(info[1]), lm_iterations=int(info[2]))


def PoseOptimizationSE3(pose12, intr5, obs7):
    """Optimizer::PoseOptimization(Frame*) (vision only, reference src/Optimizer.cc:3749-3978) with host buffers.
    obs7 [n,7] = Xw3 u v uRight invSigma2 (uRight < 0 = mono), intr5 = fx fy cx cy bf."""
    obs7 = np.ascontiguousarray(obs7, np.float64).reshape(-1, 7)
    out, fl, info = np.zeros(12, np.float32), np.zeros(max(len(obs7), 1), np.uint8), np.zeros(4)
    check(lib().viorb_pose_opt_se3(ptr(np.ascontiguousarray(pose12, np.float32)), ptr(np.ascontiguousarray(intr5, np.float32)), ptr(obs7),
                                   len(obs7), ptr(out), ptr(fl), ptr(info)))
    return dict(pose12=out, outlier=fl[:len(obs7)], n_inliers=int(info[0]), final_chi2=float(info[1]), lm_iterations=int(info[2]))


def LocalBundleAdjustmentNavState(kfs, n_local, prev_kf, preint, points, edge_idx, edge_obs, gw, cam, stop=None):
    """Optimizer::LocalBundleAdjustmentNavState (reference src/Optimizer.cc:1690-2241) with host buffers, solved on the GPU.
    kfs [NK,22] local window first; preint [W,142]; points [NP,3]; edge_idx [NE,2] int32 (point, kf) sorted by point;
    edge_obs [NE,3] = u v invSigma2; stop = optional int32 array of one element (pbStopFlag)."""
    kfs = np.ascontiguousarray(kfs, np.float64).reshape(-1, 22); preint = np.ascontiguousarray(preint, np.float64).reshape(-1, 142)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    ei = np.ascontiguousarray(edge_idx, np.int32).reshape(-1, 2); eo = np.ascontiguousarray(edge_obs, np.float64).reshape(-1, 3)
    ko, po = np.zeros((n_local, 22)), np.zeros_like(points)
    er, info = np.zeros(max(len(ei), 1), np.uint8), np.zeros(6)
    check(lib().viorb_local_ba_navstate(ptr(kfs), len(kfs), n_local, prev_kf, ptr(preint), ptr(points), len(points), ptr(ei), ptr(eo), len(ei),
                                        ptr(np.ascontiguousarray(gw, np.float64)), ptr(np.ascontiguousarray(cam, np.float64)),
                                        ptr(stop) if stop is not None else None, ptr(ko), ptr(po), ptr(er), ptr(info)))
    return dict(kfs=ko, points=po, erase=er[:len(ei)], chi2_first=info[0], chi2_final=info[1], its_first=int(info[2]), its_second=int(info[3]))


def LocalBundleAdjustmentNavStateBatch(problems, max_in_flight=128):
    """Several LocalBundleAdjustmentNavState windows solved concurrently (viorb_local_ba_navstate_batch): `problems` is a list of dicts
    with the keyword arguments of LocalBundleAdjustmentNavState (kfs, n_local, prev_kf, preint, points, edge_idx, edge_obs, gw, cam);
    returns the list of result dicts, each identical to what the single call returns."""
    n = len(problems)
    W = (capi.LbaWindow * max(n, 1))()
    keep = []
    for i, q in enumerate(problems):
        kfs = np.ascontiguousarray(q["kfs"], np.float64).reshape(-1, 22); preint = np.ascontiguousarray(q["preint"], np.float64).reshape(-1, 142)
        points = np.ascontiguousarray(q["points"], np.float64).reshape(-1, 3)
        ei = np.ascontiguousarray(q["edge_idx"], np.int32).reshape(-1, 2); eo = np.ascontiguousarray(q["edge_obs"], np.float64).reshape(-1, 3)
        gw = np.ascontiguousarray(q["gw"], np.float64); cam = np.ascontiguousarray(q["cam"], np.float64)
        ko, po = np.zeros((q["n_local"], 22)), np.zeros_like(points)
        er, info = np.zeros(max(len(ei), 1), np.uint8), np.zeros(6)
        keep.append((kfs, preint, points, ei, eo, gw, cam, ko, po, er, info))
        w = W[i]
        w.kfs = kfs.ctypes.data; w.nk = len(kfs); w.n_local = int(q["n_local"]); w.prev_kf = int(q["prev_kf"]); w.preint = preint.ctypes.data
        w.points = points.ctypes.data; w.np = len(points); w.edge_idx = ei.ctypes.data; w.edge_obs = eo.ctypes.data; w.ne = len(ei)
        w.gw = gw.ctypes.data; w.cam = cam.ctypes.data; w.stop = None
        w.kfs_out = ko.ctypes.data; w.points_out = po.ctypes.data; w.erase = er.ctypes.data; w.info = info.ctypes.data; w.status = 0
    check(lib().viorb_local_ba_navstate_batch(C.cast(W, C.c_void_p), n, int(max_in_flight)))
    return [dict(kfs=k[7], points=k[8], erase=k[9][:len(k[3])], chi2_first=k[10][0], chi2_final=k[10][1], its_first=int(k[10][2]),
                 its_second=int(k[10][3])) for k in keep]


def LocalBundleAdjustment(kfs, n_local, points, edge_idx, edge_obs, intr5, stop=None):
    """Vision-only Optimizer::LocalBundleAdjustment (reference src/Optimizer.cc:3980-4311) with host buffers, solved on the GPU.
    kfs [NK,7] = qx qy qz qw tx ty tz (Tcw), free ones first; edge_obs [NE,4] = u v uRight(<0 mono) invSigma2; intr5 = fx fy cx cy bf."""
    kfs = np.ascontiguousarray(kfs, np.float64).reshape(-1, 7); points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    ei = np.ascontiguousarray(edge_idx, np.int32).reshape(-1, 2); eo = np.ascontiguousarray(edge_obs, np.float64).reshape(-1, 4)
    ko, po = np.zeros((n_local, 7)), np.zeros_like(points)
    er, info = np.zeros(max(len(ei), 1), np.uint8), np.zeros(6)
    check(lib().viorb_local_ba_se3(ptr(kfs), len(kfs), n_local, ptr(points), len(points), ptr(ei), ptr(eo), len(ei), ptr(np.ascontiguousarray(intr5, np.float64)),
                                   ptr(stop) if stop is not None else None, ptr(ko), ptr(po), ptr(er), ptr(info)))
    return dict(kfs=ko, points=po, erase=er[:len(ei)], chi2_first=info[0], chi2_final=info[1], its_first=int(info[2]), its_second=int(info[3]))


def LocalBundleAdjustmentBatch(problems, max_in_flight=128):
    """Several vision-only LocalBundleAdjustment windows kept in flight together (viorb_local_ba_se3_batch): `problems` is a list of
    dicts with the keyword arguments of LocalBundleAdjustment (kfs, n_local, points, edge_idx, edge_obs, intr5)."""
    n = len(problems)
    W = (capi.LbaSe3Window * max(n, 1))()
    keep = []
    for i, q in enumerate(problems):
        kfs = np.ascontiguousarray(q["kfs"], np.float64).reshape(-1, 7); points = np.ascontiguousarray(q["points"], np.float64).reshape(-1, 3)
        ei = np.ascontiguousarray(q["edge_idx"], np.int32).reshape(-1, 2); eo = np.ascontiguousarray(q["edge_obs"], np.float64).reshape(-1, 4)
        intr = np.ascontiguousarray(q["intr5"], np.float64)
        ko, po = np.zeros((q["n_local"], 7)), np.zeros_like(points)
        er, info = np.zeros(max(len(ei), 1), np.uint8), np.zeros(6)
        keep.append((kfs, points, ei, eo, intr, ko, po, er, info))
        w = W[i]
        w.kfs = kfs.ctypes.data; w.nk = len(kfs); w.n_local = int(q["n_local"]); w.points = points.ctypes.data; w.np = len(points)
        w.edge_idx = ei.ctypes.data; w.edge_obs = eo.ctypes.data; w.ne = len(ei); w.intr5 = intr.ctypes.data; w.stop = None
        w.kfs_out = ko.ctypes.data; w.points_out = po.ctypes.data; w.erase = er.ctypes.data; w.info = info.ctypes.data; w.status = 0
    check(lib().viorb_local_ba_se3_batch(C.cast(W, C.c_void_p), n, int(max_in_flight)))
    return [dict(kfs=k[5], points=k[6], erase=k[7][:len(k[2])], chi2_first=k[8][0], chi2_final=k[8][1], its_first=int(k[8][2]),
                 its_second=int(k[8][3])) for k in keep]


class ORBVocabulary:
    """The part of ORBVocabulary (DBoW2::TemplatedVocabulary<FORB>, reference include/ORBVocabulary.h:30-31) the trackers use:
    transform(features, BowVector, FeatureVector, levelsup). `voc` = flat tree arrays (layout of viorb_vocabulary_create)."""

    def __init__(self, voc):
        self.L = int(voc["L"]); self.n_nodes = len(voc["word_id"])
        h = C.c_void_p()
        a = lambda k, dt: np.ascontiguousarray(voc[k], dt)
        check(lib().viorb_vocabulary_create(self.n_nodes, self.L, ptr(a("child_start", np.int32)), ptr(a("child_ids", np.int32)), ptr(a("desc", np.uint8)),
                                            ptr(a("word_id", np.int32)), ptr(a("weight", np.float64)), C.byref(h)))
        self.h = h

    @classmethod
    def load(cls, path, binary=None):
        """ORBVocabulary::loadFromTextFile / loadFromBinaryFile (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1351-1508)."""
        binary = path.endswith(".bin") if binary is None else binary
        flat = read_vocabulary_file(path, binary)
        return cls(flat)

    def close(self):
        if getattr(self, "h", None):
            lib().viorb_vocabulary_destroy(self.h); self.h = None

    __del__ = close

    def transform_features(self, desc, levelsup=4):
        """word, weight, node per descriptor (TemplatedVocabulary.h:1231-1272)."""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); n = len(desc)
        word, weight, node = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1)), np.zeros(max(n, 1), np.int32)
        check(lib().viorb_bow_transform(self.h, ptr(desc), n, levelsup, ptr(word), ptr(weight), ptr(node)))
        return word[:n], weight[:n], node[:n]

    def transform(self, desc, levelsup=4):
        """Frame::ComputeBoW: (BowVector as sorted (ids, values) with L1 normalisation, per-feature node ids with -1 for stopped
        words — the flat form of FeatureVector). TemplatedVocabulary.h:1140-1208, BowVector.cpp:36-85."""
        word, weight, node = self.transform_features(desc, levelsup)
        bow = {}
        for w, wt in zip(word.tolist(), weight.tolist()):          # addWeight in feature order
            if wt > 0:
                bow[w] = bow.get(w, 0.0) + wt
        ids = sorted(bow)
        norm = 0.0
        for k in ids:
            norm += abs(bow[k])
        vals = np.array([bow[k] / norm if norm > 0 else bow[k] for k in ids])
        return np.array(ids, np.int32), vals, np.where(weight > 0, node, -1).astype(np.int32)


def read_vocabulary_file(path, binary):
    """Parse a vocabulary file (text or binary format of the reference) into the flat-tree dict ORBVocabulary takes. Host only."""
    f = capi.VocabularyFlat()
    check(lib().viorb_vocabulary_read_file(path.encode(), int(binary), C.byref(f)))
    try:
        n = f.n_nodes
        arr = lambda p, dt, cnt: np.ctypeslib.as_array(C.cast(p, C.POINTER(dt)), shape=(cnt,)).copy()
        cs = arr(f.child_start, C.c_int32, n + 1)
        return dict(k=f.k, L=f.L, n_words=f.n_words, child_start=cs, child_ids=arr(f.child_ids, C.c_int32, int(cs[n])) if cs[n] else np.zeros(0, np.int32),
                    word_id=arr(f.word_id, C.c_int32, n), desc=arr(f.desc, C.c_uint8, 32 * n).reshape(n, 32), weight=arr(f.weight, C.c_double, n))
    finally:
        lib().viorb_vocabulary_flat_free(C.byref(f))


def write_vocabulary_file(path, voc, binary):
    """saveToTextFile / saveToBinaryFile (reference TemplatedVocabulary.h:1437-1460, :1511-1533) of a flat tree whose ids are in file order."""
    a = lambda k, dt: np.ascontiguousarray(voc[k], dt)
    fn = lib().viorb_vocabulary_save_binary if binary else lib().viorb_vocabulary_save_text
    check(fn(path.encode(), len(voc["word_id"]), int(voc.get("k", 10)), int(voc["L"]), ptr(a("child_start", np.int32)), ptr(a("child_ids", np.int32)),
             ptr(a("desc", np.uint8)), ptr(a("weight", np.float64))))


def SearchByBoW(kf_kps, kf_desc, kf_node, kf_has_point, f_kps, f_desc, f_node, nnratio=0.7, check_orientation=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (reference src/ORBmatcher.cc:159-288), host buffers.
    Returns (nmatches, match[nF]) with match = key-frame feature index or -1."""
    kk = np.ascontiguousarray(kf_kps, capi.KP_DTYPE); fk = np.ascontiguousarray(f_kps, capi.KP_DTYPE)
    m = np.full(max(len(fk), 1), -1, np.int32); n = C.c_int(0)
    check(lib().viorb_search_by_bow(ptr(kk), ptr(np.ascontiguousarray(kf_desc, np.uint8)), ptr(np.ascontiguousarray(kf_node, np.int32)),
                                    ptr(np.ascontiguousarray(kf_has_point, np.uint8)), len(kk), ptr(fk), ptr(np.ascontiguousarray(f_desc, np.uint8)),
                                    ptr(np.ascontiguousarray(f_node, np.int32)), len(fk), float(nnratio), int(check_orientation), ptr(m), C.byref(n)))
    return n.value, m[:len(fk)]


def SearchForTriangulation(k1, d1, has_point1, uright1, node1, k2, d2, has_point2, uright2, node2, F12, Cw1, pose12_2, intr4, scale_factors2,
                           level_sigma2_2, only_stereo=False, check_orientation=True):
    """ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:657-823), host buffers. Returns (nmatches, match12[N1])."""
    k1 = np.ascontiguousarray(k1, capi.KP_DTYPE); k2 = np.ascontiguousarray(k2, capi.KP_DTYPE)
    u8 = lambda a: np.ascontiguousarray(a, np.uint8); f32 = lambda a: np.ascontiguousarray(a, np.float32); i32 = lambda a: np.ascontiguousarray(a, np.int32)
    sf = f32(scale_factors2)
    m = np.full(max(len(k1), 1), -1, np.int32); n = C.c_int(0)
    check(lib().viorb_search_for_triangulation(ptr(k1), ptr(u8(d1)), ptr(u8(has_point1)), ptr(f32(uright1)), ptr(i32(node1)), len(k1), ptr(k2), ptr(u8(d2)),
                                               ptr(u8(has_point2)), ptr(f32(uright2)), ptr(i32(node2)), len(k2), ptr(f32(F12)), ptr(f32(Cw1)), ptr(f32(pose12_2)),
                                               ptr(f32(intr4)), ptr(sf), ptr(f32(level_sigma2_2)), len(sf), int(only_stereo), int(check_orientation), ptr(m),
                                               C.byref(n)))
    return n.value, m[:len(k1)]


def Fuse(kps, desc, uright, bounds, pose12, intr5, scale_factors, inv_level_sigma2, pts_f, pts_valid, pts_desc, th=3.0):
    """ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (reference src/ORBmatcher.cc:825-975), host buffers: (nFused, best_idx[npts])."""
    kps = np.ascontiguousarray(kps, capi.KP_DTYPE)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    pts_f = f32(pts_f).reshape(-1, 8); sf = f32(scale_factors)
    bi = np.full(max(len(pts_f), 1), -1, np.int32); n = C.c_int(0)
    check(lib().viorb_fuse(ptr(kps), ptr(np.ascontiguousarray(desc, np.uint8)), ptr(f32(uright)), len(kps), ptr(f32(bounds)), ptr(f32(pose12)), ptr(f32(intr5)),
                           ptr(sf), ptr(f32(inv_level_sigma2)), len(sf), ptr(pts_f), ptr(np.ascontiguousarray(pts_valid, np.uint8)),
                           ptr(np.ascontiguousarray(pts_desc, np.uint8)), len(pts_f), float(th), ptr(bi), C.byref(n)))
    return n.value, bi[:len(pts_f)]


class Frontend:
    """Batched device-resident front-end: every method only enqueues kernels on the given torch stream."""

    def __init__(self, cam, gw, scale_factors, inv_level_sigma2, bounds=(0.0, 752.0, 0.0, 480.0), max_batch=1, cap=1016,
                 check_orientation=True, device=0, dist_coef=None):
        self.L = lib()
        cfg = capi.FrontendConfig()
        if dist_coef is not None:
            for i, v in enumerate(list(dist_coef)[:5]):
                cfg.dist_coef[i] = float(np.float32(v))
        cfg.min_x, cfg.max_x, cfg.min_y, cfg.max_y = [float(b) for b in bounds]
        cfg.fx, cfg.fy, cfg.cx, cfg.cy = [float(np.float32(v)) for v in cam[:4]]
        for i in range(16):
            cfg.cam[i] = float(cam[i])
        for i in range(3):
            cfg.gravity[i] = float(gw[i])
        nl = len(scale_factors)
        for i in range(16):
            cfg.scale_factors[i] = float(scale_factors[min(i, nl - 1)])
            cfg.inv_level_sigma2[i] = float(inv_level_sigma2[min(i, nl - 1)])
        cfg.nlevels = nl
        cfg.check_orientation = int(check_orientation)
        h = C.c_void_p()
        check(self.L.viorb_frontend_create(C.byref(cfg), max_batch, cap, device, C.byref(h)))
        self.h, self.cap, self.max_batch, self.cfg = h, cap, max_batch, cfg

    def __del__(self):
        if getattr(self, "h", None):
            self.L.viorb_frontend_destroy(self.h)
            self.h = None

    @staticmethod
    def _st(stream):
        import torch
        st = stream if stream is not None else torch.cuda.current_stream()
        return C.c_void_p(st.cuda_stream)

    def undistort(self, kps_ptr, count_ptr, batch, kps_un, stream=None):
        """Frame::UndistortKeyPoints for the batch: kps_un [batch, cap] keypoint records (uint8 view [batch, cap, 28])."""
        check(self.L.viorb_frontend_undistort_device(self.h, C.c_void_p(kps_ptr), C.c_void_p(count_ptr), batch, ptr(kps_un), self._st(stream)))

    def grid(self, kps_ptr, count_ptr, batch, cell_start, cell_idx, stream=None):
        check(self.L.viorb_frontend_grid_device(self.h, C.c_void_p(kps_ptr), C.c_void_p(count_ptr), batch, ptr(cell_start),
                                                ptr(cell_idx), self._st(stream)))

    def imu_predict(self, imu, t_last, t_cur, last_ns, preint, cur_ns, pose12, stream=None):
        B, n = imu.shape[0], imu.shape[1]
        check(self.L.viorb_frontend_imu_predict_device(self.h, ptr(imu), n, ptr(t_last), ptr(t_cur), ptr(last_ns), B, ptr(preint),
                                                       ptr(cur_ns), ptr(pose12), self._st(stream)))

    def search_projection(self, cur_kps_ptr, cur_desc_ptr, cur_count_ptr, cell_start, cell_idx, pose12, last_kps, last_count,
                          last_flags, last_Pw, last_desc, th, batch, cur_match, nmatches, status, stream=None, retry_below=0):
        """retry_below > 0: only the streams whose nmatches is below it are searched again (TrackWithIMU's 2*th retry)."""
        p = lambda a: a if isinstance(a, C.c_void_p) else (C.c_void_p(a) if isinstance(a, int) else ptr(a))
        check(self.L.viorb_frontend_search_projection_retry_device(
            self.h, p(cur_kps_ptr), p(cur_desc_ptr), p(cur_count_ptr), ptr(cell_start), ptr(cell_idx), ptr(pose12), p(last_kps),
            p(last_count), ptr(last_flags), ptr(last_Pw), p(last_desc), float(th), int(retry_below), batch, ptr(cur_match), ptr(nmatches),
            ptr(status), self._st(stream)))

    def search_local_points(self, cur_kps_ptr, cur_desc_ptr, cur_count_ptr, cell_start, cell_idx, pose12, pts_f, pts_flags, pts_desc,
                            pts_count, th, nnratio, cur_owner_obs, batch, match, nmatches, frustum, status, stream=None, cur_uright=None, bf=0.0,
                            frustum_xr=None):
        p = lambda a: a if isinstance(a, C.c_void_p) else (C.c_void_p(a) if isinstance(a, int) else ptr(a))
        if cur_uright is not None:                # stereo / RGB-D frame: the mvuRight gate of ORBmatcher.cc:91-97
            check(self.L.viorb_frontend_search_local_points_stereo_device(
                self.h, p(cur_kps_ptr), p(cur_desc_ptr), p(cur_count_ptr), ptr(cur_uright), float(bf), ptr(cell_start), ptr(cell_idx), ptr(pose12), ptr(pts_f),
                ptr(pts_flags), ptr(pts_desc), ptr(pts_count), pts_f.shape[1], float(th), float(nnratio), ptr(cur_owner_obs), batch,
                ptr(match), ptr(nmatches), ptr(frustum) if frustum is not None else None, ptr(frustum_xr) if frustum_xr is not None else None, ptr(status),
                self._st(stream)))
            return
        check(self.L.viorb_frontend_search_local_points_device(
            self.h, p(cur_kps_ptr), p(cur_desc_ptr), p(cur_count_ptr), ptr(cell_start), ptr(cell_idx), ptr(pose12), ptr(pts_f),
            ptr(pts_flags), ptr(pts_desc), ptr(pts_count), pts_f.shape[1], float(th), float(nnratio), ptr(cur_owner_obs), batch,
            ptr(match), ptr(nmatches), ptr(frustum) if frustum is not None else None, ptr(status), self._st(stream)))

    def build_observations(self, kps_ptr, count_ptr, match, match_Pw, batch, obs, obs_index, n_obs, stream=None):
        p = lambda a: a if isinstance(a, C.c_void_p) else (C.c_void_p(a) if isinstance(a, int) else ptr(a))
        check(self.L.viorb_frontend_build_observations_device(self.h, p(kps_ptr), p(count_ptr), ptr(match), ptr(match_Pw), batch,
                                                              ptr(obs), ptr(obs_index), ptr(n_obs), self._st(stream)))

    def discard_outliers(self, match, obs_index, outlier, n_obs, pt_flags, batch, owner_obs, n_map, stream=None):
        """TrackWithIMU's "Discard outliers" loop (reference src/Tracking.cc:489-507)."""
        check(self.L.viorb_frontend_discard_outliers_device(self.h, ptr(match), ptr(obs_index), ptr(outlier), ptr(n_obs), ptr(pt_flags), batch,
                                                            ptr(owner_obs), ptr(n_map), self._st(stream)))

    def pose_from_navstate(self, ns, batch, pose12, stream=None):
        """Frame::UpdatePoseFromNS (reference src/Frame.cc:88-105)."""
        check(self.L.viorb_frontend_pose_from_navstate_device(self.h, ptr(ns), batch, ptr(pose12), self._st(stream)))

    def build_observations2(self, kps_ptr, count_ptr, match_a, Pw_a, match_b, pts_b, batch, obs, obs_index, n_obs, stream=None):
        p = lambda a: a if isinstance(a, C.c_void_p) else (C.c_void_p(a) if isinstance(a, int) else ptr(a))
        check(self.L.viorb_frontend_build_observations2_device(self.h, p(kps_ptr), p(count_ptr), ptr(match_a), ptr(Pw_a), ptr(match_b), ptr(pts_b),
                                                               pts_b.shape[1], batch, ptr(obs), ptr(obs_index), ptr(n_obs), self._st(stream)))

    def synth_local_points(self, kps_ptr, count_ptr, pose12_true, Pw, batch, pts_f, stream=None):
        p = lambda a: a if isinstance(a, C.c_void_p) else (C.c_void_p(a) if isinstance(a, int) else ptr(a))
        check(self.L.viorb_synth_local_points_device(self.h, p(kps_ptr), p(count_ptr), ptr(pose12_true), ptr(Pw), batch, ptr(pts_f), self._st(stream)))

    def pose_opt_se3(self, pose12, obs7, n_obs, bf, batch, out_pose12, outlier, info, stream=None):
        check(self.L.viorb_frontend_pose_opt_se3_device(self.h, ptr(pose12), ptr(obs7), ptr(n_obs), float(bf), batch, ptr(out_pose12),
                                                        ptr(outlier), ptr(info), self._st(stream)))

    def pose_opt(self, variant, compute_marg, cur_ns, last_ns, prior_ns, marg_cov_inv, preint, obs_cur, n_cur, obs_last, n_last,
                 batch, out_ns, out_last_ns, outlier_cur, outlier_last, marg_out, info, stream=None):
        check(self.L.viorb_frontend_pose_opt_device(
            self.h, variant, int(compute_marg), ptr(cur_ns), ptr(last_ns), ptr(prior_ns), ptr(marg_cov_inv), ptr(preint), ptr(obs_cur),
            ptr(n_cur), ptr(obs_last), ptr(n_last), batch, ptr(out_ns), ptr(out_last_ns), ptr(outlier_cur), ptr(outlier_last),
            ptr(marg_out), ptr(info), self._st(stream)))

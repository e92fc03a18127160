"""Oracle-side twin of viorb_amd.tracker.BatchedTracker for ONE stream — TEST INFRASTRUCTURE ONLY (used by
tests/ and by bench.py's cpu_baseline leg). Same call order as Tracking::TrackWithIMU's steady state
(reference src/Tracking.cc:412-534): extract -> grid -> pre-integrate/predict -> SearchByProjection(th=15)
-> PoseOptimization(Frame, Frame, marg), and with track_local_map=True the steady state of TrackLocalMapWithIMU
(reference src/Tracking.cc:228-346) after it: discard outliers -> SearchLocalPoints -> PoseOptimization(Frame, Frame, marg)."""
import numpy as np
from . import binding as ora
from viorb_amd import synth


class OracleTracker:
    LOCAL_FRAMES = 2

    def __init__(self, cam, gw, width=752, height=480, nfeatures=1000, th=15.0, compute_marg=True, track_local_map=False, dist_coef=None):
        self.ex = ora.Extractor(nfeatures, 1.2, 8, 20, 7)
        self.tab = self.ex.tables()
        self.cam, self.gw, self.th = np.asarray(cam, np.float64), np.asarray(gw, np.float64), float(th)
        # Frame::ComputeImageBounds / UndistortKeyPoints (Frame.cc:584-644): mK and mDistCoef are float matrices
        self.K4 = np.asarray(cam[:4], np.float32)
        self.dist = np.zeros(5, np.float32) if dist_coef is None else np.asarray(dist_coef, np.float32)
        self.bounds = tuple(float(v) for v in ora.image_bounds(width, height, self.K4, self.dist))
        self.compute_marg = compute_marg
        self.track_local_map = track_local_map
        self.local = []                      # newest first: (pts_f, flags, desc) of the frames before the last one

    def _extract(self, image):
        """ExtractORB + UndistortKeyPoints (Frame.cc:168-171): everything downstream reads mvKeysUn."""
        k, d = self.ex(image)
        if self.dist[0] != 0 and len(k):
            un = ora.undistort_points(np.stack([k["x"], k["y"]], 1), self.K4, self.dist)
            k = k.copy(); k["x"] = un[:, 0]; k["y"] = un[:, 1]
        return k, d

    def _adopt(self, kps, desc, pose_true, ns, t):
        if self.track_local_map and hasattr(self, "last_pts_f"):
            self.local = [(self.last_pts_f, self.last_flags, self.last_desc)] + self.local[:self.LOCAL_FRAMES - 1]
        self.last_kps, self.last_desc = kps, desc
        self.last_Pw = synth.plane_points_f32(np.stack([kps["x"], kps["y"]], 1), pose_true, self.cam)
        self.last_flags = np.full(len(kps), 1 | 4, np.uint8)
        self.last_ns, self.prior_ns, self.t_last = ns.copy(), ns.copy(), float(t)
        if self.track_local_map:
            self.last_pts_f = synth.local_points_f32(kps["octave"], pose_true, self.last_Pw, self.tab["scale"])

    def bootstrap(self, image, pose_true, t0, ns0, marg_cov_inv):
        k, d = self._extract(image)
        self.marg_cov_inv = np.asarray(marg_cov_inv, np.float64).reshape(12, 12).copy()
        self._adopt(k, d, pose_true, np.asarray(ns0, np.float64), t0)

    def step(self, image, imu, t_cur, pose_true, t_next_last=None, reset_ns=None, reset_marg=None, map_updated=False, recent_reloc=False,
             last_points=None):
        """One frame. Returns a dict with the intermediate results and `state`: 0 ok, 1 nmatches < 20 (TrackWithIMU returns false before
        optimising, Tracking.cc:446-447), 2 nmatchesMap < 10 (revert, :518-533), 3 mnMatchesInliers < 15 (revert, :333-342), 4 recent
        relocalisation and mnMatchesInliers < 30 (false without revert, :330-331). map_updated selects PoseOptimization(Frame, KeyFrame)
        with the last frame as the key frame it was just promoted to (:454, :243). last_points = (Pw, flags, pts_f) overrides the synthetic
        map points the NEW last frame gets (tests of the failure paths)."""
        kps, desc = self._extract(image)
        last = self.last_ns
        pre = ora.preintegrate(imu, last[10:13], last[13:16], self.t_last, t_cur)
        cur_ns = ora.predict_navstate(last, pre, self.gw)
        pose12 = ora.pose_from_navstate(cur_ns, self.cam)
        nm, match = ora.search_by_projection_frame(kps, desc, self.bounds, pose12, self.cam[:4], self.tab["scale"], self.last_flags,
                                                   self.last_Pw, self.last_desc, self.last_kps["octave"], self.last_kps["angle"], self.th)
        if nm < 20:                                          # Tracking.cc:440-444: wider window when few matches
            nm, match = ora.search_by_projection_frame(kps, desc, self.bounds, pose12, self.cam[:4], self.tab["scale"], self.last_flags,
                                                       self.last_Pw, self.last_desc, self.last_kps["octave"], self.last_kps["angle"], 2 * self.th)
        sel = np.nonzero(match >= 0)[0]
        inv_s2 = self.tab["inv_sigma2"]
        obs_cur = np.concatenate([self.last_Pw[match[sel]].astype(np.float64),
                                  np.stack([kps["x"][sel], kps["y"][sel]], 1).astype(np.float64),
                                  inv_s2[kps["octave"][sel]].astype(np.float64)[:, None]], 1).reshape(-1, 6)
        lk = self.last_kps
        has = (self.last_flags & 1) != 0                  # the last frame's own edges: keypoints that hold a map point
        obs_last = np.concatenate([self.last_Pw[has].astype(np.float64), np.stack([lk["x"][has], lk["y"][has]], 1).astype(np.float64),
                                   inv_s2[lk["octave"][has]].astype(np.float64)[:, None]], 1).reshape(-1, 6)
        tlm = self.track_local_map

        def solve(ns0, obs, marg):
            if map_updated:
                return ora.pose_opt_vi_kf(ns0, last, pre, self.gw, self.cam, obs, marg=marg)
            return ora.pose_opt_vi_frame(ns0, last, self.prior_ns, self.marg_cov_inv, pre, self.gw, self.cam, obs, obs_last, marg=marg)

        out = dict(n_kps=len(kps), nmatches=nm, match=match, pred_ns=cur_ns, kps=kps, desc=desc, state=0)
        state, final_ns, r = 0, cur_ns, None
        if nm < 20:
            state = 1
        else:
            r = solve(cur_ns, obs_cur, self.compute_marg and not tlm)
            out.update(n_inliers=r["n_inliers"], final_chi2=r["final_chi2"], ns=r["ns"], outlier_cur=r["outlier_cur"], lm_iterations=r["lm_iterations"])
            # discard outliers (Tracking.cc:489-507)
            match2 = match.copy()
            match2[sel[r["outlier_cur"][:len(sel)] != 0]] = -1
            owner = ((match2 >= 0) & ((self.last_flags[np.maximum(match2, 0)] & 4) != 0)).astype(np.uint8)
            n_map = int(owner.sum())
            out.update(n_map=n_map, match_after_discard=match2)
            if n_map < 10:
                state = 2                                   # revert: the frame keeps the IMU prediction
            else:
                final_ns = r["ns"]
        if state == 0 and tlm:
            # SearchLocalPoints + the second pose solve (:228-346)
            ns1 = final_ns
            pts_f = np.concatenate([l[0] for l in self.local]) if self.local else np.zeros((0, 8), np.float32)
            pflags = np.concatenate([l[1] for l in self.local]) if self.local else np.zeros(0, np.uint8)
            pdesc = np.concatenate([l[2] for l in self.local]) if self.local else np.zeros((0, 32), np.uint8)
            offs = np.cumsum([0] + [len(l[0]) for l in self.local])
            pose12_b = ora.pose_from_navstate(ns1, self.cam)
            log_sf = np.float32(np.log(np.float64(self.tab["scale"][1])))
            if len(pts_f):
                n_loc, loc_match, _ = ora.search_local_points(kps, desc, self.bounds, pose12_b, self.cam[:4], self.tab["scale"], log_sf, pts_f, pflags,
                                                              pdesc, 1.0, 0.8, owner)
            else:
                n_loc, loc_match = 0, np.full(len(kps), -1, np.int32)
            use_a = match2 >= 0
            use_b = (~use_a) & (loc_match >= 0)
            sel2 = np.nonzero(use_a | use_b)[0]
            X = np.where(use_a[sel2, None], self.last_Pw[np.maximum(match2[sel2], 0)], pts_f[np.maximum(loc_match[sel2], 0), :3] if len(pts_f) else 0.0)
            obs_cur2 = np.concatenate([X.astype(np.float64), np.stack([kps["x"][sel2], kps["y"][sel2]], 1).astype(np.float64),
                                       inv_s2[kps["octave"][sel2]].astype(np.float64)[:, None]], 1).reshape(-1, 6)
            r2 = solve(ns1, obs_cur2, self.compute_marg)
            # mnMatchesInliers (:307-325): inlier edges whose map point has observations
            pf = np.where(use_a[sel2], self.last_flags[np.maximum(match2[sel2], 0)], pflags[np.maximum(loc_match[sel2], 0)] if len(pflags) else 0)
            inl = int(((r2["outlier_cur"][:len(sel2)] == 0) & ((pf & 4) != 0)).sum())
            out.update(n_loc=n_loc, loc_match=loc_match, loc_offsets=offs, n_inliers2=r2["n_inliers"], final_chi2_2=r2["final_chi2"], ns2=r2["ns"],
                       n_obs2=len(sel2), inliers=inl, lm_iterations2=r2["lm_iterations"])
            if recent_reloc and inl < 30:
                state, final_ns, r = 4, r2["ns"], r2
            elif inl < 15:
                state = 3                                   # revert to the state this stage started from
            else:
                final_ns, r = r2["ns"], r2
        out["state"] = state
        out["final_ns"] = final_ns
        if self.compute_marg and state in (0, 4) and r is not None:
            self.marg_cov_inv = r["marg_cov_inv"].copy()
        t_adopt = t_cur if t_next_last is None else t_next_last
        if reset_ns is not None:                       # key-frame boundary of the harness: the next frame starts from the given state / prior
            if reset_marg is not None:
                self.marg_cov_inv = np.asarray(reset_marg, np.float64).reshape(12, 12).copy()
            self._adopt(kps, desc, pose_true, np.asarray(reset_ns, np.float64), t_adopt)
        else:
            self._adopt(kps, desc, pose_true, final_ns, t_adopt)
        if last_points is not None:
            self.last_Pw, self.last_flags = np.asarray(last_points[0], np.float32).copy(), np.asarray(last_points[1], np.uint8).copy()
            if self.track_local_map and last_points[2] is not None:
                self.last_pts_f = np.asarray(last_points[2], np.float32).copy()
        return out

"""Oracle-side twin of viorb_amd.tracker.BatchedTracker for ONE stream — TEST INFRASTRUCTURE ONLY (used by
tests/ and by bench.py's cpu_baseline leg). Same call order as Tracking::TrackWithIMU's steady state
(reference src/Tracking.cc:412-534): extract -> grid -> pre-integrate/predict -> SearchByProjection(th=15)
-> PoseOptimization(Frame, Frame, marg)."""
import numpy as np
from . import binding as ora
from viorb_amd import synth


class OracleTracker:
    def __init__(self, cam, gw, width=752, height=480, nfeatures=1000, th=15.0, compute_marg=True):
        self.ex = ora.Extractor(nfeatures, 1.2, 8, 20, 7)
        self.tab = self.ex.tables()
        self.cam, self.gw, self.th = np.asarray(cam, np.float64), np.asarray(gw, np.float64), float(th)
        self.bounds = (0.0, float(width), 0.0, float(height))
        self.compute_marg = compute_marg

    def _adopt(self, kps, desc, pose_true, ns, t):
        self.last_kps, self.last_desc = kps, desc
        self.last_Pw = synth.plane_points_f32(np.stack([kps["x"], kps["y"]], 1), pose_true, self.cam)
        self.last_flags = np.full(len(kps), 1 | 4, np.uint8)
        self.last_ns, self.prior_ns, self.t_last = ns.copy(), ns.copy(), float(t)

    def bootstrap(self, image, pose_true, t0, ns0, marg_cov_inv):
        k, d = self.ex(image)
        self.marg_cov_inv = np.asarray(marg_cov_inv, np.float64).reshape(12, 12).copy()
        self._adopt(k, d, pose_true, np.asarray(ns0, np.float64), t0)

    def step(self, image, imu, t_cur, pose_true, t_next_last=None):
        kps, desc = self.ex(image)
        last = self.last_ns
        pre = ora.preintegrate(imu, last[10:13], last[13:16], self.t_last, t_cur)
        cur_ns = ora.predict_navstate(last, pre, self.gw)
        pose12 = ora.pose_from_navstate(cur_ns, self.cam)
        nm, match = ora.search_by_projection_frame(kps, desc, self.bounds, pose12, self.cam[:4], self.tab["scale"], self.last_flags,
                                                   self.last_Pw, self.last_desc, self.last_kps["octave"], self.last_kps["angle"], self.th)
        sel = np.nonzero(match >= 0)[0]
        inv_s2 = self.tab["inv_sigma2"]
        obs_cur = np.concatenate([self.last_Pw[match[sel]].astype(np.float64),
                                  np.stack([kps["x"][sel], kps["y"][sel]], 1).astype(np.float64),
                                  inv_s2[kps["octave"][sel]].astype(np.float64)[:, None]], 1).reshape(-1, 6)
        lk = self.last_kps
        obs_last = np.concatenate([self.last_Pw.astype(np.float64), np.stack([lk["x"], lk["y"]], 1).astype(np.float64),
                                   inv_s2[lk["octave"]].astype(np.float64)[:, None]], 1).reshape(-1, 6)
        r = ora.pose_opt_vi_frame(cur_ns, last, self.prior_ns, self.marg_cov_inv, pre, self.gw, self.cam, obs_cur, obs_last,
                                  marg=self.compute_marg)
        out = dict(n_kps=len(kps), nmatches=nm, match=match, n_inliers=r["n_inliers"], final_chi2=r["final_chi2"], ns=r["ns"],
                   pred_ns=cur_ns, outlier_cur=r["outlier_cur"], kps=kps, desc=desc)
        if self.compute_marg:
            self.marg_cov_inv = r["marg_cov_inv"].copy()
        self._adopt(kps, desc, pose_true, r["ns"], t_cur if t_next_last is None else t_next_last)
        return out

"""Batched per-frame tracking sequence on the device — the build's counterpart of the call order in
Tracking::TrackWithIMU (reference src/Tracking.cc:412-534) for B independent mono-inertial streams:

    extract -> AssignFeaturesToGrid -> IMU pre-integration + NavState prediction (PredictNavStateByIMU)
            -> SearchByProjection(cur, last, th=15) -> PoseOptimization(cur, last frame, preint, gw, marg)
    and, with track_local_map=True, the steady state of Tracking::TrackLocalMapWithIMU (reference src/Tracking.cc:228-346) after it:
            -> discard outliers -> SearchLocalPoints (isInFrustum + SearchByProjection(F, local points, th=1), nnratio 0.8)
            -> PoseOptimization(cur, last frame, preint, gw, marg=true) on the last-frame and local-map matches together
    (stage one then runs without the marginal, as the reference does). The local map is the points created for the
    LOCAL_FRAMES frames before the last one (UpdateLocalMap itself is map management, outside the hot path).

Everything stays in HBM; the only host work per step is enqueueing kernels. Two HIP streams: extraction of
frame k+1 (bandwidth/ALU-bound, fills the chip) overlaps the matching + pose solve of frame k (latency-bound,
one workgroup per stream); two extractor handles alternate so frame k's keypoints stay valid meanwhile. Map maintenance (creating
map points for the new last frame) is NOT part of the reference's per-frame path (LocalMapping does it);
here it is supplied by the synthetic plane world of viorb_amd/synth.py through
viorb_synth_plane_points_device, using the stream's ground-truth pose of that frame.
torch tensors only carry device memory and the stream."""
import ctypes as C
import time
import numpy as np
from .capi import lib, check, ptr, KP_DTYPE
from . import capi as capi_mod
from .extractor import ORBextractor
from .frontend import Frontend
from . import synth


class BatchedTracker:
    MAX_STEPS_AHEAD = 8
    LOCAL_FRAMES = 2

    def __init__(self, cam, gw, batch, width=752, height=480, nfeatures=1000, th=15.0, device=0, compute_marg=True, overlap=True,
                 track_local_map=False):
        import torch
        self.torch = torch
        self.B, self.w, self.h, self.th = batch, width, height, float(th)
        self.dev = torch.device("cuda", device)
        self.exs = [ORBextractor(nfeatures, 1.2, 8, 20, 7, max_batch=batch, device=device) for _ in range(2 if overlap else 1)]
        self.ex = self.exs[0]
        t = self.ex.tables()
        self.cap = self.ex.capacity_for(width, height)
        self.overlap = overlap
        self.s_ex = torch.cuda.Stream(device=self.dev) if overlap else None
        self.s_tr = torch.cuda.Stream(device=self.dev) if overlap else None
        self.ev_ex = [torch.cuda.Event() for _ in range(2)]
        self.ev_tr = [None, None]
        self._in_flight = []
        self.k = 0
        self.k_rolls = 0
        self.fe = Frontend(cam, gw, t["scale"], t["inv_sigma2"], (0.0, float(width), 0.0, float(height)), max_batch=batch,
                           cap=self.cap, device=device)
        self.cam = np.asarray(cam, np.float64)
        self.compute_marg = compute_marg
        B, cap = batch, self.cap
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.dev)
        # last-frame state
        self.last_kps = z((B, cap, KP_DTYPE.itemsize), torch.uint8)
        self.last_desc = z((B, cap, 32), torch.uint8)
        self.last_count = z((B,), torch.int32)
        self.last_flags = z((B, cap), torch.uint8)
        self.last_Pw = z((B, cap, 3), torch.float32)
        self.last_ns = z((B, 22), torch.float64)
        self.prior_ns = z((B, 22), torch.float64)
        self.marg_cov_inv = z((B, 144), torch.float64)
        self.t_last = z((B,), torch.float64)
        # per-step scratch / outputs
        self.cell_start = z((B, 64 * 48 + 1), torch.int32)
        self.cell_idx = z((B, cap), torch.int32)
        self.preint = z((B, 142), torch.float64)
        self.cur_ns = z((B, 22), torch.float64)
        self.pose12 = z((B, 12), torch.float32)
        self.cur_match = z((B, cap), torch.int32)
        self.nmatches = z((B,), torch.int32)
        self.status = z((B,), torch.int32)
        self.obs_cur = z((B, cap, 6), torch.float64)
        self.obs_last = z((B, cap, 6), torch.float64)
        self.idx_cur = z((B, cap), torch.int32)
        self.idx_last = z((B, cap), torch.int32)
        self.n_cur = z((B,), torch.int32)
        self.n_last = z((B,), torch.int32)
        self.last_self = z((B, cap), torch.int32)          # identity where the last keypoint has a map point
        self.out_ns = z((B, 22), torch.float64)
        self.out_last_ns = z((B, 22), torch.float64)
        self.outlier_cur = z((B, cap), torch.uint8)
        self.outlier_last = z((B, cap), torch.uint8)
        self.marg_out = z((B, 144), torch.float64)
        self.info = z((B, 4), torch.float64)
        self.iota = torch.arange(cap, dtype=torch.int32, device=self.dev)[None, :].expand(B, cap).contiguous()
        self.track_local_map = track_local_map
        if track_local_map:
            R = self.LOCAL_FRAMES
            self.last_pts_f = z((B, cap, 8), torch.float32)              # isInFrustum fields of the last frame's points
            self.loc_pts_f = z((B, R * cap, 8), torch.float32)           # local map: slot r = points created r + 1 frames before the last
            self.loc_flags = z((B, R * cap), torch.uint8)
            self.loc_desc = z((B, R * cap, 32), torch.uint8)
            self.loc_count = torch.full((B,), R * cap, dtype=torch.int32, device=self.dev)
            self.owner_obs = z((B, cap), torch.uint8)
            self.n_map = z((B,), torch.int32)
            self.pose12_b = z((B, 12), torch.float32)
            self.loc_match = z((B, cap), torch.int32)
            self.n_loc = z((B,), torch.int32)
            self.status2 = z((B,), torch.int32)
            self.obs_cur2 = z((B, cap, 6), torch.float64)
            self.idx_cur2 = z((B, cap), torch.int32)
            self.n_cur2 = z((B,), torch.int32)
            self.out_ns2 = z((B, 22), torch.float64)
            self.outlier_cur2 = z((B, cap), torch.uint8)
            self.info2 = z((B, 4), torch.float64)

    # -- helpers ------------------------------------------------------------------------------------
    def _cur_ptrs(self):
        return self.ex.results_device()            # kps, desc, count, status, cap

    def _roll(self, true_pose12, t_cur, ns_for_last, marg_src=None, stream=None):
        """Make the frame just processed the new last frame (one fused kernel: local-map shift, keypoint / descriptor / state copies)
        and give its keypoints map points. Runs on the current torch stream (the tracking stream when overlapping)."""
        kps, desc, count, _, cap = self._cur_ptrs()
        L = lib()
        st = Frontend._st(stream)
        tlm = self.track_local_map
        vp = lambda tns: C.c_void_p(tns.data_ptr()) if tns is not None else None
        check(L.viorb_frontend_roll_device(
            self.fe.h, C.c_void_p(kps), C.c_void_p(desc), C.c_void_p(count), vp(self.last_kps), vp(self.last_desc), vp(self.last_count),
            vp(self.last_pts_f) if tlm else None, vp(self.last_flags), vp(self.loc_pts_f) if tlm else None, vp(self.loc_desc) if tlm else None,
            vp(self.loc_flags) if tlm else None, self.LOCAL_FRAMES, int(tlm and self.k_rolls > 0), vp(ns_for_last), vp(self.last_ns), vp(self.prior_ns),
            vp(t_cur), vp(self.t_last), vp(marg_src), vp(self.marg_cov_inv) if marg_src is not None else None, self.B, st))
        check(L.viorb_synth_plane_points_device(self.fe.h, vp(self.last_kps), vp(self.last_count), ptr(true_pose12), synth.PLANE_Z0, self.B,
                                                ptr(self.last_Pw), ptr(self.last_flags), vp(self.last_self), st))
        if tlm:
            self.fe.synth_local_points(self.last_kps.data_ptr(), self.last_count.data_ptr(), true_pose12, self.last_Pw, self.B, self.last_pts_f)
        self.k_rolls += 1

    def bootstrap(self, images, true_pose12, t0, ns0, marg_cov_inv):
        """First frame of every stream: extract, adopt as last frame with ground-truth state."""
        self.torch.cuda.synchronize()
        self.k_rolls = 0
        self.ex = self.exs[0]
        self.ex.extract_batch_device(images)
        self.marg_cov_inv.copy_(marg_cov_inv)
        self._roll(true_pose12, t0, ns0)
        self.torch.cuda.synchronize()
        self.k = 0
        self.ev_tr = [None, None]

    def _track_pre(self, imu, t_cur):
        """The part of a tracking step that does not need the new frame's keypoints: IMU pre-integration + prediction and the last
        frame's own observations. With overlap it is enqueued before the wait for the extraction."""
        B = self.B
        self.fe.imu_predict(imu, self.t_last, t_cur, self.last_ns, self.preint, self.cur_ns, self.pose12)
        self.fe.build_observations(self.last_kps.data_ptr(), self.last_count.data_ptr(), self.last_self, self.last_Pw, B, self.obs_last,
                                   self.idx_last, self.n_last)

    def _track(self, imu, t_cur, true_pose12, chain_estimate, true_ns, t_next_last, marg_reset=None):
        B = self.B
        kps, desc, count, _, cap = self._cur_ptrs()
        fe = self.fe
        fe.grid(kps, count, B, self.cell_start, self.cell_idx)
        fe.search_projection(kps, desc, count, self.cell_start, self.cell_idx, self.pose12, self.last_kps.data_ptr(),
                             self.last_count.data_ptr(), self.last_flags, self.last_Pw, self.last_desc.data_ptr(), self.th, B,
                             self.cur_match, self.nmatches, self.status)
        # "if(nmatches<20) ... SearchByProjection(..., 2*th, ...)" (reference src/Tracking.cc:440-444): a no-op for streams with >= 20 matches
        fe.search_projection(kps, desc, count, self.cell_start, self.cell_idx, self.pose12, self.last_kps.data_ptr(),
                             self.last_count.data_ptr(), self.last_flags, self.last_Pw, self.last_desc.data_ptr(), 2 * self.th, B,
                             self.cur_match, self.nmatches, self.status, retry_below=20)
        fe.build_observations(kps, count, self.cur_match, self.last_Pw, B, self.obs_cur, self.idx_cur, self.n_cur)
        tlm = self.track_local_map
        fe.pose_opt(1, self.compute_marg and not tlm, self.cur_ns, self.last_ns, self.prior_ns, self.marg_cov_inv, self.preint, self.obs_cur,
                    self.n_cur, self.obs_last, self.n_last, B, self.out_ns, self.out_last_ns, self.outlier_cur, self.outlier_last, self.marg_out,
                    self.info)
        final_ns = self.out_ns
        if tlm:
            # ---- TrackLocalMapWithIMU (reference src/Tracking.cc:228-346)
            fe.discard_outliers(self.cur_match, self.idx_cur, self.outlier_cur, self.n_cur, self.last_flags, B, self.owner_obs, self.n_map)
            fe.pose_from_navstate(self.out_ns, B, self.pose12_b)
            fe.search_local_points(kps, desc, count, self.cell_start, self.cell_idx, self.pose12_b, self.loc_pts_f, self.loc_flags, self.loc_desc,
                                   self.loc_count, 1.0, 0.8, self.owner_obs, B, self.loc_match, self.n_loc, None, self.status2)
            fe.build_observations2(kps, count, self.cur_match, self.last_Pw, self.loc_match, self.loc_pts_f, B, self.obs_cur2, self.idx_cur2,
                                   self.n_cur2)
            fe.pose_opt(1, self.compute_marg, self.out_ns, self.last_ns, self.prior_ns, self.marg_cov_inv, self.preint, self.obs_cur2, self.n_cur2,
                        self.obs_last, self.n_last, B, self.out_ns2, self.out_last_ns, self.outlier_cur2, self.outlier_last, self.marg_out,
                        self.info2)
            final_ns = self.out_ns2
        self._roll(true_pose12, t_cur if t_next_last is None else t_next_last, final_ns if chain_estimate else true_ns,
                   marg_src=self.marg_out if (self.compute_marg and chain_estimate) else (marg_reset if not chain_estimate else None))

    def step(self, images, imu, t_cur, true_pose12, chain_estimate=True, true_ns=None, t_next_last=None, marg_reset=None):
        """One tracking step for all streams. images [B,h,w] u8, imu [B,n,7] f64, t_cur [B] f64,
        true_pose12 [B,12] f64 (only used to create map points for the next step). t_next_last overrides the
        stamp the frame gets as "last frame" (periodic streams: the loop-closing frame restarts at 0).
        With overlap the call returns after enqueueing; results (info, nmatches, out_ns, ...) are those of this
        step once the device is synchronised. chain_estimate=False + true_ns (+ marg_reset [B,144]) is the harness's key-frame
        boundary: the frame is processed in full, but the state (and prior information) the NEXT frame starts from is the given one —
        the reference never chains its frame-to-frame prior for long either, every key frame / map update restarts it
        (src/Tracking.cc:241-287)."""
        torch = self.torch
        if not self.overlap:
            self.ex.extract_batch_device(images)
            self._track_pre(imu, t_cur)
            self._track(imu, t_cur, true_pose12, chain_estimate, true_ns, t_next_last, marg_reset)
            return
        slot = self.k % 2
        ex = self.exs[slot]
        cur = torch.cuda.current_stream(self.dev)
        self.s_ex.wait_stream(cur)                       # inputs produced on the caller's stream
        if self.ev_tr[slot] is not None:
            self.s_ex.wait_event(self.ev_tr[slot])       # this handle's previous results have been consumed
        ex.extract_batch_device(images, stream=self.s_ex)
        self.ev_ex[slot].record(self.s_ex)
        self.s_tr.wait_stream(cur)
        with torch.cuda.stream(self.s_tr):
            self._track_pre(imu, t_cur)                  # runs while the extraction of this frame is still in flight
        self.s_tr.wait_event(self.ev_ex[slot])
        self.ex = ex
        with torch.cuda.stream(self.s_tr):
            self._track(imu, t_cur, true_pose12, chain_estimate, true_ns, t_next_last, marg_reset)
            ev = torch.cuda.Event(); ev.record(self.s_tr)
            self.ev_tr[slot] = ev
        self.k += 1
        # keep the host at most MAX_STEPS_AHEAD steps in front of the device: a live system never queues more (frames arrive one at a
        # time), and an unbounded backlog only holds events and command-queue slots (no measurable effect on the step time either way)
        # the inputs are read on s_ex / s_tr, not on the caller's stream: a caller that drops a tensor right after step() would hand its
        # memory back to torch's caching allocator while the GPU still reads it, so the step's inputs stay referenced until its event
        self._in_flight.append((ev, (images, imu, t_cur, true_pose12, true_ns, t_next_last, marg_reset)))
        if len(self._in_flight) > self.MAX_STEPS_AHEAD:
            self._in_flight.pop(0)[0].synchronize()

    def status_all(self):
        """Per-stream status of the last step folded over the extractor and both searches (0 = ok, VIORB_ERR_CAPACITY when a level's
        quadtree input or a search candidate list was truncated): assert on it — a truncated stream no longer equals the reference."""
        torch = self.torch
        _, _, _, ex_status, _ = self._cur_ptrs()
        st = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
        check(_hip_memcpy_dtod_async(st.data_ptr(), ex_status, 4 * self.B, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
        torch.cuda.synchronize()
        out = st.cpu().numpy()
        for other in (self.status, getattr(self, "status2", None)):
            if other is not None:
                o = other.cpu().numpy()
                out = np.where(out == 0, o, out)
        return out


def _hip_memcpy_dtod_async(dst, src, nbytes, stream):
    """Device-to-device copy of raw device addresses on `stream`, through libviorb_hip's own HIP runtime binding
    (dlopen-ing libamdhip64 by name from Python could load a second runtime next to torch's)."""
    return lib().viorb_memcpy_dtod_async(C.c_void_p(dst), C.c_void_p(src), nbytes, stream)


class NativeTracker:
    """The C++ batched tracking sequence (viorb_tracker_*, viorb_amd/csrc/tracker.hip): one host call per frame enqueues
    TrackWithIMU + TrackLocalMapWithIMU for B streams with the reference's thresholds and backup / revert decisions taken per stream
    on the device (reference src/Tracking.cc:229-346, 412-534). This class only marshals torch tensors into the C structs and keeps the
    inputs of the steps still in flight alive (they are read on the tracker's own HIP streams: a caller that drops a tensor right after
    step() would otherwise hand its memory back to torch's caching allocator while the GPU still reads it)."""
    STATE_NAMES = ("ok", "few_matches", "revert_1", "revert_2", "reloc_few")

    def __init__(self, cam, gw, batch, width=752, height=480, nfeatures=1000, th=15.0, device=0, compute_marg=True, track_local_map=True,
                 local_frames=2, max_steps_ahead=8, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, dist_coef=None):
        import torch
        self.torch = torch
        self.B, self.w, self.h = batch, width, height
        self.dev = torch.device("cuda", device)
        cfg = capi_mod.TrackerConfig()
        if dist_coef is not None:                     # Frame::mDistCoef = k1 k2 p1 p2 [k3]: keypoints are undistorted ahead of the grid
            for i, v in enumerate(list(dist_coef)[:5]):
                cfg.frontend.dist_coef[i] = float(np.float32(v))
        cfg.extractor.nfeatures, cfg.extractor.scale_factor, cfg.extractor.nlevels = nfeatures, scale_factor, nlevels
        cfg.extractor.ini_th_fast, cfg.extractor.min_th_fast = ini_th, min_th
        fe = cfg.frontend
        fe.fx, fe.fy, fe.cx, fe.cy = [float(np.float32(v)) for v in cam[:4]]
        for i in range(16):
            fe.cam[i] = float(cam[i])
        for i in range(3):
            fe.gravity[i] = float(gw[i])
        fe.check_orientation = 1
        cfg.width, cfg.height, cfg.batch, cfg.device = width, height, batch, device
        cfg.th_projection = float(th); cfg.track_local_map = int(track_local_map); cfg.local_frames = int(local_frames)
        cfg.compute_marg = int(compute_marg); cfg.max_steps_ahead = int(max_steps_ahead); cfg.synth_plane_z0 = float(synth.PLANE_Z0)
        h = C.c_void_p()
        check(lib().viorb_tracker_create(C.byref(cfg), C.byref(h)))
        self.hnd = h
        cap = C.c_int()
        check(lib().viorb_tracker_capacity(h, C.byref(cap)))
        self.cap = cap.value
        self.track_local_map = int(track_local_map) > 0
        self._keep, self._pending = [], []
        self.max_steps_ahead = max_steps_ahead

    def close(self):
        if getattr(self, "hnd", None):
            lib().viorb_tracker_destroy(self.hnd); self.hnd = None

    __del__ = close

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def bootstrap(self, images, true_pose12, t0, ns0, marg_cov_inv):
        assert images.is_cuda and images.dim() == 3 and images.stride(2) == 1
        check(lib().viorb_tracker_bootstrap(self.hnd, ptr(images), images.stride(1), images.stride(0), ptr(ns0), ptr(t0), ptr(marg_cov_inv),
                                            ptr(true_pose12), self._stream()))
        self._keep, self._pending = [], []

    def step(self, images, imu, t_cur, true_pose12=None, map_updated=None, recent_reloc=None, t_next_last=None, reset_ns=None, reset_marg=None):
        """images [B,h,w] u8, imu [B,n,7] f64, t_cur [B] f64 — CUDA tensors. true_pose12 [B,12] f64: map points of the new last frame from
        the synthetic plane world (None: call set_last_points before the next step). map_updated / recent_reloc: [B] u8 flags.
        `images` may also be a HOST tensor (page-locked: torch.Tensor.pin_memory()): the live-feed path, uploaded by the tracker's copy stream."""
        inp = capi_mod.TrackerInputs()
        if images.is_cuda:
            inp.d_images = images.data_ptr()
        else:
            inp.h_images = images.data_ptr()
        inp.image_stride = images.stride(1); inp.image_pitch_bytes = images.stride(0)
        inp.d_imu = imu.data_ptr(); inp.n_imu = imu.shape[1]; inp.d_t_cur = t_cur.data_ptr()
        opt = lambda t: t.data_ptr() if t is not None else None
        inp.d_map_updated = opt(map_updated); inp.d_recent_reloc = opt(recent_reloc); inp.d_t_next_last = opt(t_next_last)
        inp.d_reset_ns = opt(reset_ns); inp.d_reset_marg = opt(reset_marg); inp.d_synth_pose12 = opt(true_pose12)
        check(lib().viorb_tracker_step(self.hnd, C.byref(inp), self._stream()))
        # ONE record per step: its inputs plus whatever set_last_points handed over since the previous step (those copies are
        # enqueued ahead of this step on the tracking stream, so they have completed once this step's event has). The C++ throttle returns
        # only when at most `ahead` steps are still in flight (max_steps_ahead <= 0 means 8 there), i.e. every record older than that has
        # been consumed.
        self._keep.append((images, imu, t_cur, true_pose12, map_updated, recent_reloc, t_next_last, reset_ns, reset_marg, self._pending))
        self._pending = []
        ahead = self.max_steps_ahead if self.max_steps_ahead > 0 else 8
        while len(self._keep) > ahead + 1:
            self._keep.pop(0)

    def set_last_points(self, Pw, flags, pts_f=None):
        check(lib().viorb_tracker_set_last_points_device(self.hnd, ptr(Pw), ptr(flags), ptr(pts_f) if pts_f is not None else None, self._stream()))
        self._pending.append((Pw, flags, pts_f))

    def sync(self):
        check(lib().viorb_tracker_sync(self.hnd))
        self._keep, self._pending = [], []

    def host_stats(self, reset=False):
        a, b, n = C.c_double(), C.c_double(), C.c_longlong()
        check(lib().viorb_tracker_host_stats(self.hnd, C.byref(a), C.byref(b), C.byref(n), int(reset)))
        return dict(enqueue_s=a.value, throttle_s=b.value, steps=n.value)

    def results(self, names=None):
        """Synchronise and copy results of the last step to the host: dict of numpy arrays."""
        self.sync()
        r = capi_mod.TrackerResults()
        check(lib().viorb_tracker_results_device(self.hnd, C.byref(r)))
        B, cap = self.B, self.cap
        shapes = dict(state=((B,), np.int32), status=((B,), np.int32), nmatches=((B,), np.int32), n_map=((B,), np.int32), n_loc=((B,), np.int32),
                      inliers=((B,), np.int32), n_obs=((B,), np.int32), n_obs2=((B,), np.int32), cur_match=((B, cap), np.int32),
                      loc_match=((B, cap), np.int32), last_count=((B,), np.int32), info=((B, 4), np.float64), info2=((B, 4), np.float64),
                      pred_ns=((B, 22), np.float64), ns_stage1=((B, 22), np.float64), ns_stage2=((B, 22), np.float64),
                      final_ns=((B, 22), np.float64), final_marg=((B, 144), np.float64), last_ns=((B, 22), np.float64),
                      outlier_cur=((B, cap), np.uint8), outlier_cur2=((B, cap), np.uint8), last_flags=((B, cap), np.uint8),
                      last_Pw=((B, cap, 3), np.float32), last_pts_f=((B, cap, 8), np.float32))
        out = {}
        for k in (names or shapes.keys()):
            shp, dt = shapes[k]
            a = np.zeros(shp, dt)
            check(lib().viorb_memcpy_dtoh(ptr(a), C.c_void_p(getattr(r, k)), a.nbytes))
            out[k] = a
        return out

    def device_ptrs(self):
        r = capi_mod.TrackerResults()
        check(lib().viorb_tracker_results_device(self.hnd, C.byref(r)))
        return r


class DropinTracker:
    """ONE stream through the host-buffer drop-ins, call by call — what a VIORB `Tracking` thread pays per frame when it is built with the
    shims of viorb_amd/shim/ (INTEGRATION.md): viorb_extract, viorb_preintegrate, viorb_search_by_projection_frame, viorb_pose_opt_vi,
    viorb_search_by_projection_points, viorb_pose_opt_vi. The glue between the calls (NavState prediction, Frame::UpdatePoseFromNS, edge
    lists, discarding outliers, the synthetic map of viorb_amd/synth.py) is the caller's host code, written in numpy here. Same sequence as
    the batched tracker (TrackWithIMU + TrackLocalMapWithIMU, reference src/Tracking.cc:412-534, 228-346). `times` accumulates the wall
    time spent inside each drop-in call."""
    LOCAL_FRAMES = 2

    def __init__(self, cam, gw, width=752, height=480, nfeatures=1000, th=15.0, dist_coef=None):
        from . import frontend as fe_mod
        from .extractor import ORBextractor
        self.fe = fe_mod
        self.ex = ORBextractor(nfeatures, 1.2, 8, 20, 7)
        self.tab = self.ex.tables()
        self.cam, self.gw, self.th = np.asarray(cam, np.float64), np.asarray(gw, np.float64), float(th)
        self.K4 = np.asarray(cam[:4], np.float32)
        self.dist = np.zeros(5, np.float32) if dist_coef is None else np.asarray(dist_coef, np.float32)
        self.bounds = tuple(float(v) for v in fe_mod.ComputeImageBounds(width, height, self.K4, self.dist))      # Frame::ComputeImageBounds, once
        self.matcher = fe_mod.ORBmatcher(0.9, True)
        self.local = []
        self.times = {}

    def _extract(self, image):
        """Frame::ExtractORB + Frame::UndistortKeyPoints: everything downstream reads mvKeysUn."""
        k, d = self._timed("viorb_extract", self.ex, image)
        if self.dist[0] != 0 and len(k):
            un = self._timed("viorb_undistort_points", self.fe.UndistortKeyPoints, np.stack([k["x"], k["y"]], 1), self.K4, self.dist)
            k = k.copy(); k["x"] = un[:, 0]; k["y"] = un[:, 1]
        return k, d

    def _timed(self, name, f, *a, **kw):
        t0 = time.perf_counter()
        r = f(*a, **kw)
        d = self.times.setdefault(name, [0.0, 0])
        d[0] += time.perf_counter() - t0; d[1] += 1
        return r

    @staticmethod
    def _qmat(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    @staticmethod
    def _mat_q(R):
        from scipy.spatial.transform import Rotation
        return Rotation.from_matrix(R).as_quat()

    def _predict(self, last, pre):
        """Converter::updateNS (reference src/Converter.cc:27-49) + Frame::SetInitialNavStateAndBias on the caller's side."""
        ns = last.copy()
        ns[10:13] = last[10:13] + last[16:19]; ns[13:16] = last[13:16] + last[19:22]; ns[16:22] = 0
        dP, dV, dR, dt = pre[0:3], pre[3:6], pre[6:15].reshape(3, 3), pre[141]
        Rw = self._qmat(last[6:10])
        ns[0:3] = last[0:3] + last[3:6] * dt + 0.5 * self.gw * dt * dt + Rw @ dP
        ns[3:6] = last[3:6] + self.gw * dt + Rw @ dV
        ns[6:10] = self._mat_q(Rw @ dR)
        return ns

    def _pose12(self, ns):
        Rcw, tcw = synth.cam_pose_from_navstate(ns, self.cam)
        return np.concatenate([Rcw.ravel(), tcw]).astype(np.float32)

    def _adopt(self, kps, desc, pose_true, ns, t):
        if hasattr(self, "last_pts_f"):
            self.local = [(self.last_pts_f, self.last_flags, self.last_desc)] + self.local[:self.LOCAL_FRAMES - 1]
        self.last_kps, self.last_desc = kps, desc
        self.last_Pw = synth.plane_points_f32(np.stack([kps["x"], kps["y"]], 1), pose_true, self.cam)
        self.last_flags = np.full(len(kps), 1 | 4, np.uint8)
        self.last_ns, self.prior_ns, self.t_last = ns.copy(), ns.copy(), float(t)
        self.last_pts_f = synth.local_points_f32(kps["octave"], pose_true, self.last_Pw, self.tab["scale"])

    def bootstrap(self, image, pose_true, t0, ns0, marg_cov_inv):
        k, d = self._extract(image)
        self.marg_cov_inv = np.asarray(marg_cov_inv, np.float64).reshape(12, 12).copy()
        self._adopt(k, d, pose_true, np.asarray(ns0, np.float64), t0)

    def step(self, image, imu, t_cur, pose_true, t_next_last=None, reset_ns=None, reset_marg=None, map_updated=False):
        fe = self.fe
        kps, desc = self._extract(image)
        last = self.last_ns
        pre = self._timed("viorb_preintegrate", fe.preintegrate, imu, last[10:13], last[13:16], self.t_last, t_cur)
        cur_ns = self._predict(last, pre)
        pose12 = self._pose12(cur_ns)
        search = lambda th: self._timed("viorb_search_by_projection_frame", self.matcher.SearchByProjection, kps, desc, self.bounds, pose12, self.cam[:4],
                                        self.tab["scale"], self.last_kps, self.last_flags, self.last_Pw, self.last_desc, th)
        nm, match = search(self.th)
        if nm < 20:
            nm, match = search(2 * self.th)
        state, final_ns, marg = 0, cur_ns, None
        inv_s2 = self.tab["inv_sigma2"]
        edges = lambda P, k, sel: np.concatenate([P.astype(np.float64), np.stack([k["x"][sel], k["y"][sel]], 1).astype(np.float64),
                                                  inv_s2[k["octave"][sel]].astype(np.float64)[:, None]], 1).reshape(-1, 6)
        lk = self.last_kps
        has = (self.last_flags & 1) != 0
        obs_last = edges(self.last_Pw[has], lk, has)

        def solve(ns0, obs, want_marg):
            if map_updated:
                return self._timed("viorb_pose_opt_vi", fe.PoseOptimization, ns0, last, pre, self.gw, self.cam, obs, last_is_keyframe=True, bComputeMarg=want_marg)
            return self._timed("viorb_pose_opt_vi", fe.PoseOptimization, ns0, last, pre, self.gw, self.cam, obs, obs_last, self.prior_ns, self.marg_cov_inv,
                               last_is_keyframe=False, bComputeMarg=want_marg)
        n_inl = 0
        if nm < 20:
            state = 1
        else:
            sel = np.nonzero(match >= 0)[0]
            r = solve(cur_ns, edges(self.last_Pw[match[sel]], kps, sel), False)
            match2 = match.copy(); match2[sel[r["outlier_cur"][:len(sel)] != 0]] = -1
            owner = ((match2 >= 0) & ((self.last_flags[np.maximum(match2, 0)] & 4) != 0)).astype(np.uint8)
            if int(owner.sum()) < 10:
                state = 2
            else:
                ns1 = r["ns"]
                pts_f = np.concatenate([l[0] for l in self.local]) if self.local else np.zeros((0, 8), np.float32)
                pflags = np.concatenate([l[1] for l in self.local]) if self.local else np.zeros(0, np.uint8)
                pdesc = np.concatenate([l[2] for l in self.local]) if self.local else np.zeros((0, 32), np.uint8)
                loc_match = np.full(len(kps), -1, np.int32)
                if len(pts_f):
                    _, loc_match = self._timed("viorb_search_by_projection_points", fe.SearchLocalPoints, kps, desc, self.bounds, self._pose12(ns1), self.cam[:4],
                                               self.tab["scale"], pts_f, pflags, pdesc, 1.0, 0.8, owner)
                use_a = match2 >= 0
                sel2 = np.nonzero(use_a | (loc_match >= 0))[0]
                X = np.where(use_a[sel2, None], self.last_Pw[np.maximum(match2[sel2], 0)], pts_f[np.maximum(loc_match[sel2], 0), :3] if len(pts_f) else 0.0)
                r2 = solve(ns1, edges(X, kps, sel2), True)
                pf = np.where(use_a[sel2], self.last_flags[np.maximum(match2[sel2], 0)], pflags[np.maximum(loc_match[sel2], 0)] if len(pflags) else 0)
                n_inl = int(((r2["outlier_cur"][:len(sel2)] == 0) & ((pf & 4) != 0)).sum())
                if n_inl < 15:
                    state, final_ns = 3, ns1
                else:
                    final_ns, marg = r2["ns"], r2["marg_cov_inv"]
        if marg is not None:
            self.marg_cov_inv = marg.copy()
        t_adopt = t_cur if t_next_last is None else t_next_last
        if reset_ns is not None:
            if reset_marg is not None:
                self.marg_cov_inv = np.asarray(reset_marg, np.float64).reshape(12, 12).copy()
            self._adopt(kps, desc, pose_true, np.asarray(reset_ns, np.float64), t_adopt)
        else:
            self._adopt(kps, desc, pose_true, final_ns, t_adopt)
        return dict(state=state, nmatches=nm, inliers=n_inl, final_ns=final_ns)

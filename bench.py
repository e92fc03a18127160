#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: frames/s of the per-frame visual-inertial front-end (ORB extract + match +
PoseOptimization), plus the other BASELINE configs behind --config.

    python bench.py --gpus N --steps K --warmup W [--config euroc|synth720p|kitti_stereo|local_ba]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

--config euroc (default, BASELINE configs[1]): a "step" is one new 752x480 frame for each of the S independent mono-inertial streams a
GPU owns, through the C++ batched tracker (viorb_tracker_step: TrackWithIMU + TrackLocalMapWithIMU with the reference's thresholds and
revert decisions per stream on the device). --config synth720p (configs[4]): the same sequence at 1280x720 / 1500 features, 8 streams
per GPU. --config kitti_stereo (configs[2]): a step is P 1241x376 stereo pairs: extraction of both images (2000 features each) +
Frame::ComputeStereoMatches. --config local_ba (configs[3]): a step is one batch of W = 20 LocalBundleAdjustmentNavState windows.
Weak scaling: every rank owns the same amount of work; the path shards by stream / pair / window with no data-path collective, the
only RCCL traffic is the (units, time) reduction at the end. Inputs are resident in HBM before the timed region (local_ba: host
buffers, as its caller is the LocalMapping thread). Rank 0 prints ONE JSON line. `roofline` / `roofline_pose` are measured live with
HIP events on the launching stream; `cpu_baseline` times the CPU oracle (a port: the reference cannot be built here) on the host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

N_FRAMES = 8                                  # frames per (periodic) synthetic stream
HBM_PEAK_GBS = 8000.0                         # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6                # MI355X FP64 vector spec (AMD data sheet: 256 CUs x 128 FLOP/clk x 2.4 GHz; the guide has no FP64 row)
CONFIGS = {
    # name: width, height, features, default units per GPU, BASELINE.json configs index
    "euroc": dict(w=752, h=480, nfeat=1000, streams=1024, baseline_config=1),      # 1024 streams per GPU per step: two pose-solver workgroups per CU (512: 172 k, 768: 173 k, 1024: 177 k, 1536: 163 k frames/s)
    "synth720p": dict(w=1280, h=720, nfeat=1500, streams=8, baseline_config=4),
    "kitti_stereo": dict(w=1241, h=376, nfeat=2000, streams=256, baseline_config=2),
    "local_ba": dict(w=752, h=480, nfeat=1000, streams=256, baseline_config=3),
    # one stream through the host-buffer drop-ins, call by call: the reference's own calling pattern (a Tracking thread built with viorb_amd/shim/)
    "dropin": dict(w=752, h=480, nfeat=1000, streams=1, baseline_config=1),
}
EXTRACT_KERNELS = ("k_copy_level0", "k_resize", "k_fast_cells", "k_octree", "k_octree_large", "k_octree_big", "k_blur", "k_orient_describe")   # k_undistort is Frame::UndistortKeyPoints: the tracking side


def level_pixels(w, h, nlevels=8, sf=1.2):
    """Sum of level pixels P: level l = cvRound(w / sf^l) x cvRound(h / sf^l), float scale table as the reference."""
    scale, tot, sizes = np.float32(1.0), 0, []
    for l in range(nlevels):
        inv = np.float32(1.0) / scale
        lw, lh = int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))
        sizes.append((lw, lh)); tot += lw * lh
        scale = np.float32(scale * np.float32(sf))
    return tot, sizes


def algo_bytes(w, h, nfeat):
    """Algorithmic bytes per image of each extractor kernel (B_ext = 4 P + K (709 + 961 + 60))."""
    P, _ = level_pixels(w, h)
    return {"k_fast_cells": P, "k_blur": 2 * P, "k_resize": P - w * h, "k_copy_level0": w * h, "k_orient_describe": nfeat * (709 + 961 + 60)}, P


def load_traffic_table():
    """HBM bytes per launch of the extractor / solver kernels from the PMC counters (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
    separate passes, 2 x FETCH_SIZE + WRITE_SIZE as calibrated by tools/ubench/fetch_calib.hip), recorded per round under profiles/."""
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):         # the newest round's table that exists
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            pass
    return None


def _gen_stream(args):
    seed, w, h, dist = args
    from viorb_amd.synth import make_periodic_stream
    s = make_periodic_stream(seed, N_FRAMES, w, h, dist=dist)
    return dict(frames=s["frames"], imu=s["imu"], t=s["t"], ns_true=s["ns_true"], pose_true=s["pose_true"], period=s["period"], cam=s["cam"], gw=s["gw"])


def generator_procs(n_seeds, procs=None):
    """Worker processes of the synthetic-stream generator on this rank: the host's cores are shared by the ranks of the node (LOCAL_WORLD_SIZE)."""
    nproc = max(1, min(n_seeds, (os.cpu_count() or 2) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))), 16))
    return max(1, min(nproc, procs)) if procs else nproc


def host_memory_estimate(cfg, streams, distinct, world, host_pass):
    """Bytes of host memory one rank of a tracking config holds at its peak: the generated distinct streams, their [F, distinct, h, w] stack that
    is uploaded once (the replication to S streams happens on the device), and (only when a live-feed pass reads it: N = 1 or --host-input) the
    page-locked copy of all S streams' frames."""
    frame = cfg["w"] * cfg["h"]
    return {"generated": distinct * N_FRAMES * frame, "upload_stack": distinct * N_FRAMES * frame, "page_locked_copy": streams * N_FRAMES * frame if host_pass else 0}


def generate_streams(seeds, w=752, h=480, procs=None, dist=None):
    """CPU-side synthetic data (before anything touches the GPU). `procs=1` generates in-process: under `rocprofv3 --pmc` the profiler's
    preloaded library has initialised the GPU before Python starts, and forked pool workers of such a process never exit (that, not a
    kernel-ordering bug, is why the round-2 bench "did not finish" under --pmc: profiles/README.md, round 3)."""
    import multiprocessing as mp
    nproc = generator_procs(len(seeds), procs)
    jobs = [(s, w, h, dist) for s in seeds]
    if nproc == 1:
        return [_gen_stream(j) for j in jobs]
    with mp.get_context("fork").Pool(nproc) as pool:
        return pool.map(_gen_stream, jobs)


# ------------------------------------------------------------------------------------------------------------------------------
# cpu_baseline legs (the oracle is the checker / baseline only; never on the product path)
# ------------------------------------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _oracle_track_frames(args):
    """One oracle stream: per-frame wall times of `n` frames after `warm` warm-up frames (the example mains' convention,
    reference Examples/Stereo/stereo_kitti.cc:80-124)."""
    s, w, h, nfeat, warm, n = args[:6]
    dist = args[6] if len(args) > 6 else None
    from oracle.harness import OracleTracker
    tr = OracleTracker(s["cam"], s["gw"], w, h, nfeat, track_local_map=True, dist_coef=dist)
    mci = np.eye(12) * 1e3
    tr.bootstrap(s["frames"][0], s["pose_true"][0], s["t"][0], s["ns_true"][0], mci)
    times = []
    for k in range(1, warm + n + 1):
        j = k % N_FRAMES
        t0 = time.perf_counter()
        if j:
            tr.step(s["frames"][j], s["imu"][j], s["t"][j], s["pose_true"][j], map_updated=(j == 1 and k > 1))
        else:
            tr.step(s["frames"][0], s["imu"][0], s["period"], s["pose_true"][0], t_next_last=0.0, reset_ns=s["ns_true"][0], reset_marg=mci)
        dt = time.perf_counter() - t0
        if k > warm:
            times.append(dt)
    return times


def cpu_baseline_tracking(streams, w, h, nfeat, frames_1core, warm, dist=None):
    import multiprocessing as mp
    hw = os.cpu_count() or 1
    t1 = _oracle_track_frames((streams[0], w, h, nfeat, warm, frames_1core, dist))
    one = dict(value=round(len(t1) / sum(t1), 3), median_ms=round(float(np.median(t1)) * 1e3, 3), mean_ms=round(float(np.mean(t1)) * 1e3, 3), frames=len(t1))
    # all cores: one oracle stream per hardware thread (the batched counterpart), each frames_1core / 4 frames after the warm-up
    per = max(10, frames_1core // 4)
    nproc = min(hw, len(streams))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(nproc) as pool:
        res = pool.map(_oracle_track_frames, [(streams[i], w, h, nfeat, min(warm, 4), per, dist) for i in range(nproc)])
    wall = time.perf_counter() - t0
    busy = sum(sum(r) for r in res)
    allc = dict(value=round(nproc * per / (busy / nproc), 3), cores=nproc, frames=nproc * per, wall_s=round(wall, 2),
                note="sum of timed frames / mean per-process busy time (process start and warm-up frames excluded)")
    return {"value": one["value"], "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames after %d warm-up frames of one synthetic stream, same per-frame sequence (oracle, C++ -O3, 1 host thread): "
                      "median %.2f ms, mean %.2f ms per frame" % (one["frames"], warm, one["median_ms"], one["mean_ms"]),
            "median_ms_per_frame": one["median_ms"], "mean_ms_per_frame": one["mean_ms"], "all_cores": allc,
            "hardware_concurrency": hw, "cpu_model": cpu_model()}


# ------------------------------------------------------------------------------------------------------------------------------
# tracking configs (euroc, synth720p)
# ------------------------------------------------------------------------------------------------------------------------------
def run_tracking(args, cfg, rank, dev_index, dev, world):
    import ctypes as C
    import torch
    import torch.distributed as dist
    import viorb_amd
    from viorb_amd.distributed import stream_seeds
    from viorb_amd.tracker import NativeTracker
    W_IMG, H_IMG, NFEAT, S = cfg["w"], cfg["h"], cfg["nfeat"], args.streams
    TLM = not args.no_track_local_map
    # at most 256 distinct synthetic streams are generated per rank (CPU time); beyond that the streams repeat (own copy of the images, independent
    # tracker state each)
    distinct = min(S, args.distinct or 256)
    # the EuRoC camera of the reference's settings file is distorted (Examples/ROS/ORB_VIO/launch/euroc.yaml:64-67): the synthetic frames are rendered
    # through that lens and the tracker undistorts the keypoints ahead of the grid (Frame::UndistortKeyPoints) with bounds from the undistorted corners
    from viorb_amd.synth import EUROC_DIST
    lens = EUROC_DIST if (args.config == "euroc" and not args.no_distortion) else None
    # the synthetic streams come from main(), generated before the process group and the GPU were touched (a pool forked after RCCL / HIP
    # initialisation is not something to rely on); keyed by the camera model so that the pinhole pass finds its own
    pre = getattr(args, "pregenerated", {}).get("lens" if lens else "pinhole")
    base = pre if pre is not None else generate_streams(stream_seeds(rank, distinct), W_IMG, H_IMG, args.gen_procs, lens)
    streams = [base[i % distinct] for i in range(S)]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # the distinct streams' frames go up once and are replicated ON THE DEVICE (every stream its own copy of the images): the host never holds the
    # [F, S, h, w] stack (3 GB at 1024 streams; 8 ranks share one host)
    frames_d = up(np.stack([s["frames"] for s in base[:distinct]], 1))       # [F, distinct, h, w] u8
    frames = frames_d[:, torch.arange(S, device=dev) % distinct].contiguous() if S != distinct else frames_d   # [F, S, h, w] u8
    del frames_d
    imu = up(np.stack([s["imu"] for s in streams], 1))                         # [F, S, n, 7] f64
    t_frames = up(np.stack([s["t"] for s in streams], 1))                      # [F, S]
    t_period = up(np.array([s["period"] for s in streams]))                    # [S]
    pose_true = up(np.stack([s["pose_true"] for s in streams], 1))             # [F, S, 12] f64
    ns_true = up(np.stack([s["ns_true"] for s in streams], 1))                 # [F, S, 22]
    zeros_t = torch.zeros(S, dtype=torch.float64, device=dev)
    mci0 = up(np.stack([np.eye(12).ravel() * 1e3] * S))
    ones_u8 = torch.ones(S, dtype=torch.uint8, device=dev)
    cam, gw = streams[0]["cam"], streams[0]["gw"]
    tr = NativeTracker(cam, gw, S, W_IMG, H_IMG, NFEAT, th=15.0, device=dev_index, compute_marg=True, track_local_map=TLM, dist_coef=lens)
    tr.bootstrap(frames[0], pose_true[0], t_frames[0], ns_true[0], mci0)

    # live feed: the same frames in page-locked host memory, uploaded by the tracker's copy stream inside the step (viorb_tracker_inputs.h_images)
    if world > 1:
        args.no_host_input_pass = True                                   # the context passes after the timed region are N = 1 only
    need_host = args.host_input or not args.no_host_input_pass       # 8 x S x frame bytes of page-locked memory: only when a pass reads it
    frames_host = torch.from_numpy(np.stack([s["frames"] for s in streams], 1)).pin_memory() if need_host else None
    feed = {"frames": frames_host if args.host_input else frames}

    def run_step(k):
        frames = feed["frames"]
        j = k % N_FRAMES
        if j == 0:
            # closing the loop: frame F == frame 0, its stamp is the period; next "last" stamp is 0. It is also the harness's key-frame
            # boundary: the frame is tracked in full, the next one starts from the key frame's state with a fresh prior (an endless
            # frame-to-frame prior chain is not what the reference runs: every key frame / map update restarts it, Tracking.cc:241-287)
            tr.step(frames[0], imu[0], t_period, pose_true[0], t_next_last=zeros_t, reset_ns=ns_true[0], reset_marg=mci0)
        elif j == 1 and k > 1:
            # the frame after the boundary sees mbMapUpdated: PoseOptimization(Frame, KeyFrame) in both stages (Tracking.cc:243, :454)
            tr.step(frames[1], imu[1], t_frames[1], pose_true[1], map_updated=ones_u8)
        else:
            tr.step(frames[j], imu[j], t_frames[j], pose_true[j])

    single_stream = None
    if world == 1 and rank == 0 and not args.no_host_input_pass:
        # north_star's single-stream figure, BEFORE the timed region (a single stream leaves the GPU nearly idle; measured after the saturating
        # batch the same sequence reads ~17 % lower): ONE stream per step (stream 0 of this run), device-resident frames, through
        # the same C++ tracker — (a) ORB extract + match only (extraction, undistortion, grid, IMU prediction, SearchByProjection with its
        # 2 th retry: track_local_map < 0), (b) the full per-frame sequence. A step is one frame; the host runs at most 8 frames ahead.
        try:
            ss = {}
            for key, mode in (("extract_match_frames_per_s", -1), ("full_sequence_frames_per_s", 1 if TLM else 0)):
                tr1 = NativeTracker(cam, gw, 1, W_IMG, H_IMG, NFEAT, th=15.0, device=dev_index, compute_marg=True, track_local_map=mode, dist_coef=lens)
                tr1.bootstrap(frames[0][:1], pose_true[0][:1], t_frames[0][:1], ns_true[0][:1], mci0[:1])
                def one(kk):
                    j = kk % N_FRAMES
                    if j == 0:
                        tr1.step(frames[0][:1], imu[0][:1], t_period[:1], pose_true[0][:1], t_next_last=zeros_t[:1], reset_ns=ns_true[0][:1], reset_marg=mci0[:1])
                    elif j == 1 and kk > 1 and mode >= 0:
                        tr1.step(frames[1][:1], imu[1][:1], t_frames[1][:1], pose_true[1][:1], map_updated=ones_u8[:1])
                    else:
                        tr1.step(frames[j][:1], imu[j][:1], t_frames[j][:1], pose_true[j][:1])
                kk = 1
                for _ in range(24):
                    one(kk); kk += 1
                tr1.sync(); torch.cuda.synchronize()
                n1, best = 300, 0.0
                for _ in range(2):                      # best of two passes (a single stream leaves the GPU nearly idle: its clock wanders)
                    t1 = time.perf_counter()
                    for _ in range(n1):
                        one(kk); kk += 1
                    tr1.sync(); torch.cuda.synchronize()
                    best = max(best, n1 / (time.perf_counter() - t1))
                ss[key] = round(best, 1)
                r1 = tr1.results(["state", "status", "nmatches"])
                ss[key.replace("_frames_per_s", "_matches_last_frame")] = int(r1["nmatches"][0])
                if int(r1["status"][0]) != 0:
                    ss[key + "_status"] = int(r1["status"][0])
                del tr1
            ss["note"] = ("one stream per step, frames resident in HBM, viorb_tracker_step; extract_match = extraction + undistortion + grid + IMU "
                          "prediction + SearchByProjection (no pose solve); north_star asks >= 2000 frames/s for it")
            single_stream = ss
        except Exception as e:                          # never fail the bench line over the context figure
            single_stream = {"error": str(e)[:200]}

    k = 1
    for _ in range(args.warmup):
        run_step(k); k += 1
    tr.sync(); torch.cuda.synchronize()
    L = viorb_amd.lib()
    timed = None if args.all_kernel_events else b"k_fast_cells,k_pose_opt_vi"
    L.viorb_profile_select(timed)
    L.viorb_profile_reset(); L.viorb_profile_enable(0 if args.no_kernel_events else 1)
    tr.host_stats(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(k); k += 1
    t_calls = time.perf_counter() - t0                 # host time inside the step calls (enqueue + the in-flight throttle)
    tr.sync(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    L.viorb_profile_enable(0)
    hs = tr.host_stats()
    # ---- the other input mode, outside the timed region: HBM-resident frames when the line was measured on the live feed and vice versa
    other = {}
    if not args.no_host_input_pass:
        feed["frames"] = frames if args.host_input else frames_host
        n_other = max(8, min(args.steps, 48))
        for _ in range(4):
            run_step(k); k += 1
        tr.sync(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_other):
            run_step(k); k += 1
        tr.sync(); torch.cuda.synchronize()
        dt_other = time.perf_counter() - t1
        other = {"frames_per_s": round(S * n_other / dt_other, 1), "ms_per_step": round(dt_other / n_other * 1e3, 4), "steps": n_other}
        feed["frames"] = frames_host if args.host_input else frames
    img_bytes = W_IMG * H_IMG

    # ---- sanity of the timed work (outside the timed region)
    res = tr.results(["state", "status", "nmatches", "n_loc", "info", "info2", "inliers"])
    info = res["info2"] if TLM else res["info"]
    names = C.create_string_buffer(4096); ms = (C.c_double * 64)(); calls = (C.c_int * 64)(); n = C.c_int()
    L.viorb_profile_read(names, 4096, ms, calls, 64, C.byref(n))
    prof = {nm_: (ms[i], calls[i]) for i, nm_ in enumerate(names.value.decode().split("\n")[:n.value])}
    if args.timeline:                                       # dev aid: the two streams' kernel intervals on one clock (HIP events, no tracer)
        cap = 16384
        t_a = (C.c_double * cap)(); t_b = (C.c_double * cap)(); t_s = (C.c_int * cap)(); t_n = C.c_int()
        L.viorb_profile_timeline(t_a, t_b, t_s, cap, C.byref(t_n))
        nms = names.value.decode().split("\n")
        rows = sorted((t_a[i], t_b[i], nms[t_s[i]]) for i in range(min(t_n.value, cap)))
        with open(args.timeline, "w") as f:
            for a, b, nm_ in rows[-int(args.timeline_rows):]:
                f.write("%10.1f %10.1f %8.1f %s\n" % (a * 1e3, b * 1e3, (b - a) * 1e3, nm_))

    out = dict(units=S * args.steps, elapsed=elapsed)
    if rank == 0:
        AB, P = algo_bytes(W_IMG, H_IMG, NFEAT)
        traffic = load_traffic_table()
        roof = roof_pose = None
        if "k_fast_cells" in prof and prof["k_fast_cells"][1]:
            tot_ms, ncalls = prof["k_fast_cells"]
            avg_s = tot_ms / ncalls * 1e-3
            per_launch = L.viorb_extractor_fast_launch_images(S)     # FAST goes out in sub-launches over the batch; the profiler times one of them per step, in rotation
            bpl = AB["k_fast_cells"] * per_launch
            tr_b = None
            if traffic and traffic.get("config") == args.config and "k_fast_cells" in traffic.get("bytes_per_launch", {}):
                upl = traffic["units_per_launch"]
                upl = upl["k_fast_cells"] if isinstance(upl, dict) else upl
                tr_b = int(traffic["bytes_per_launch"]["k_fast_cells"] * per_launch / upl)
            roof = {"bound": "hbm", "kernel": "k_fast_cells", "achieved": round(bpl / avg_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(bpl / avg_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": tr_b, "avg_launch_us": round(avg_s * 1e6, 2),
                    "algorithmic_bytes_per_launch": int(bpl),
                    "traffic_source": (traffic or {}).get("source") if tr_b else None,
                    "note": "dominant kernel of the HBM-bound extractor family; it is bound by integer vector issue, not by HBM (DESIGN.md §4)"}
            if args.all_kernel_events:
                ext_ms = sum(v[0] for kname, v in prof.items() if kname in EXTRACT_KERNELS)
                ext_bytes = (4 * P + NFEAT * (709 + 961 + 60)) * S * args.steps
                roof["extractor_all_kernels_GBps"] = round(ext_bytes / (ext_ms * 1e-3) / 1e9, 2)
                roof["extractor_all_kernels_frac"] = round(ext_bytes / (ext_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            roof["kernel_ms_per_step"] = {kname: round(v[0] / args.steps, 4) for kname, v in sorted(prof.items())}
            if not args.all_kernel_events:                  # one of FAST's sub-launches per step was timed
                roof["kernel_ms_per_step"]["k_fast_cells"] = round(tot_ms / args.steps * (-(-S // per_launch)), 4)
            roof["launches_per_step"] = -(-S // per_launch); roof["images_per_launch"] = per_launch
            roof["note"] += ("; avg_launch_us / achieved / frac are the launch beside the tracking stream's kernels (the two streams share "
                             "every CU); 'alone' = the same launch in an extraction-only pass after the timed region")
            try:                                            # context, outside the timed region: the extractor alone, same batch, same images
                ex_alone = viorb_amd.ORBextractor(NFEAT, 1.2, 8, 20, 7, max_batch=S, device=dev_index)
                for _ in range(2):
                    ex_alone.extract_batch_device(frames[0])
                torch.cuda.synchronize()
                L.viorb_profile_select(b"k_fast_cells"); L.viorb_profile_reset(); L.viorb_profile_enable(1)
                t_a = time.perf_counter()
                for i in range(8):
                    ex_alone.extract_batch_device(frames[i % N_FRAMES])
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t_a) / 8
                L.viorb_profile_enable(0)
                nm2 = C.create_string_buffer(4096); ms2 = (C.c_double * 64)(); cl2 = (C.c_int * 64)(); n2 = C.c_int()
                L.viorb_profile_read(nm2, 4096, ms2, cl2, 64, C.byref(n2))
                pa = {nm_: (ms2[i], cl2[i]) for i, nm_ in enumerate(nm2.value.decode().split("\n")[:n2.value])}
                if pa.get("k_fast_cells", (0, 0))[1]:
                    a_s = pa["k_fast_cells"][0] / pa["k_fast_cells"][1] * 1e-3
                    ext_bytes = (4 * P + NFEAT * (709 + 961 + 60)) * S
                    roof["alone"] = {"avg_launch_us": round(a_s * 1e6, 2), "achieved": round(bpl / a_s / 1e9, 2), "frac": round(bpl / a_s / 1e9 / HBM_PEAK_GBS, 5),
                                     "extractor_ms_per_batch": round(wall * 1e3, 3), "extractor_images_per_s": round(S / wall, 1),
                                     "extractor_algorithmic_GBps": round(ext_bytes / wall / 1e9, 1),
                                     "extractor_frac": round(ext_bytes / wall / 1e9 / HBM_PEAK_GBS, 5)}
                del ex_alone
            except Exception as e:                          # never fail the bench line over the context figure
                roof["alone"] = {"error": str(e)[:200]}
        if "k_pose_opt_vi" in prof and prof["k_pose_opt_vi"][1]:
            # FLOP model of one solve (counted from the kernel source): every evaluation of the objective costs
            # 190 FLOP per reprojection edge (projection, Huber weight, 2 x 6 camera-frame Jacobian, 20 + 6 accumulations) plus
            # 27 500 for the dense factors, their 24 x 24 assembly and the Cholesky solve of the trial; evaluations = LM iterations
            # + one initial evaluation per round (4 rounds). Edges and iterations are the measured ones of the last step.
            tot_ms, ncalls = prof["k_pose_opt_vi"]
            avg_s = tot_ms / ncalls * 1e-3
            its1, its2 = float(res["info"][:, 2].mean()), float(res["info2"][:, 2].mean()) if TLM else 0.0
            r2 = tr.results(["n_obs", "n_obs2", "last_count"])
            e1 = float(r2["n_obs"].mean() + r2["last_count"].mean()); e2 = float(r2["n_obs2"].mean() + r2["last_count"].mean())
            fl1 = (its1 + 4) * (e1 * 190 + 27500); fl2 = (its2 + 4) * (e2 * 190 + 27500) if TLM else 0.0
            flops_per_launch = (fl1 + fl2) / (2 if TLM else 1) * S
            ach = flops_per_launch / avg_s / 1e12
            roof_pose = {"bound": "fp64_vector", "kernel": "k_pose_opt_vi", "achieved": round(ach, 3), "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / FP64_VECTOR_PEAK_TFLOPS, 5), "traffic": None, "avg_launch_us": round(avg_s * 1e6, 2),
                         "flop_per_launch": int(flops_per_launch), "mean_lm_iterations": [round(its1, 2), round(its2, 2)],
                         "mean_edges": [round(e1, 1), round(e2, 1)], "problems_per_launch": S,
                         "note": "the kernel with the largest share of GPU time (profiles/*_kernel_stats.csv); one 256-thread workgroup per "
                                 "problem, latency / issue bound: the FP64 fraction is small by construction (DESIGN.md §4)"}
        out["roofline"], out["roofline_pose"] = roof, roof_pose
        tracked = int((res["state"] == 0).sum())
        out["config"] = {
            "workload": "%s-shaped synthetic mono-inertial streams %dx%d, 8 levels, %d features, per frame: extract + IMU pre-integration (10 samples) + "
                        "SearchByProjection(th=15, 2*th retry) + PoseOptimization + discard outliers%s; thresholds / reverts of the reference per stream "
                        "on the device; C++ tracker (viorb_tracker_step)" %
                        ("EuRoC" if args.config == "euroc" else "1280x720", W_IMG, H_IMG, NFEAT,
                         " + SearchLocalPoints(~2000 local points) + PoseOptimization(marg)  [TrackWithIMU + TrackLocalMapWithIMU]" if TLM else " [TrackWithIMU only]"),
            "baseline_config": cfg["baseline_config"], "keyframe_boundary_every_frames": N_FRAMES,
            "camera_model": ("EuRoC radial-tangential distortion k1 k2 p1 p2 = %s (euroc.yaml:64-67): frames rendered through the lens, keypoints undistorted on the "
                             "device (Frame::UndistortKeyPoints), image bounds from the undistorted corners" % (list(lens[:4]),)) if lens else "pinhole (no distortion)",
            "keyframe_variant": "PoseOptimization(Frame, KeyFrame) on the frame after every boundary (mbMapUpdated), (Frame, Frame) otherwise",
            "input": ("page-locked host memory, uploaded inside the timed region by the tracker's copy stream (live feed)" if args.host_input
                      else "HBM-resident frames (uploaded before the timed region)"),
            "track_local_map": TLM, "streams_per_gpu": S, "distinct_synthetic_streams_per_gpu": distinct, "frames_per_step": S * world, "solver_dtype": "f64",
            "host_enqueue_ms_per_step": round(hs["enqueue_s"] / max(hs["steps"], 1) * 1e3, 4),
            "host_throttle_wait_ms_per_step": round(hs["throttle_s"] / max(hs["steps"], 1) * 1e3, 4),
            "host_step_call_ms_per_step": round(t_calls / args.steps * 1e3, 4),
            "tracked_streams_last_step": tracked, "state_histogram_last_step": np.bincount(res["state"], minlength=5).tolist(),
            "mean_matches_last_step": round(float(res["nmatches"].mean()), 1), "mean_local_matches_last_step": round(float(res["n_loc"].mean()), 1),
            "mean_inliers_last_step": round(float(info[:, 0].mean()), 1), "status_ok": bool((res["status"] == 0).all())}
        if other:                                           # per rank (rank 0's figure): the driver's line stays the aggregate of the timed region
            host_fps = other["frames_per_s"] if not args.host_input else S * args.steps / elapsed
            out["config"]["host_input_frames_per_s"] = round(host_fps, 1)
            out["config"]["host_input_pcie_GBps"] = round(host_fps * img_bytes / 1e9, 2)
            out["config"]["host_input_note"] = ("per GPU; frames of %d bytes from page-locked host memory through viorb_tracker_inputs.h_images (copy stream, ring of "
                                                "max_steps_ahead + 2 device buffers); PCIe Gen5 x16 spec 63 GB/s" % img_bytes)
            if args.host_input:
                out["config"]["hbm_resident_frames_per_s"] = other["frames_per_s"]
        if single_stream is not None:
            out["config"]["single_stream"] = single_stream
        if not (res["status"] == 0).all():
            raise SystemExit("a stream reported a capacity / status error: %s" % res["status"])
        if args.config == "synth720p" and world == 1 and not args.no_host_input_pass and S < 128:
            # BASELINE configs[4] gives every GPU 8 of the 64 streams: a latency-bound batch that any number of GPUs runs at the same per-GPU
            # rate. What ONE GPU does with this workload when it has enough streams to fill it (the same 8 scenes replicated, own state each):
            import copy
            a2 = copy.copy(args); a2.distinct = distinct; a2.streams = 128; a2.steps = 24; a2.warmup = 6; a2.no_host_input_pass = True; a2.no_cpu_baseline = True; a2.no_kernel_events = True
            sat = run_tracking(a2, cfg, rank, dev_index, dev, world)
            out["config"]["saturated_frames_per_s"] = round(sat["units"] / sat["elapsed"], 1)
            out["config"]["saturated_note"] = "the same 1280x720 / 1500-feature sequence at 128 streams per GPU per step (the 8 scenes replicated, own tracker state each), 24 steps after the timed region"
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline_tracking(base, W_IMG, H_IMG, NFEAT, 200 if args.config == "euroc" else 60, 20 if args.config == "euroc" else 8, lens)
    out["metric"] = "frames/sec ORB extract+match+pose-opt, EuRoC 752x480, 1/2/4/8 GPUs" if args.config == "euroc" else \
        "frames/sec ORB extract+match+pose-opt, synthetic 1280x720 / 1500 features, batched streams"
    out["unit"] = "frames/s"; out["dtype"] = "u8"
    return out


# ------------------------------------------------------------------------------------------------------------------------------
# dropin: ONE stream, every library call a host-buffer drop-in with its own upload / launch / synchronise / download (what a single VIORB
# process pays per frame through the shims) — a latency figure, per call and per frame; never the headline
# ------------------------------------------------------------------------------------------------------------------------------
def run_dropin(args, cfg, rank, dev_index, dev, world):
    import torch
    from viorb_amd.distributed import stream_seeds
    from viorb_amd.tracker import DropinTracker
    from viorb_amd.synth import EUROC_DIST
    dist = None if args.no_distortion else EUROC_DIST
    s = generate_streams(stream_seeds(rank, 1), cfg["w"], cfg["h"], 1, dist)[0]
    tr = DropinTracker(s["cam"], s["gw"], cfg["w"], cfg["h"], cfg["nfeat"], dist_coef=dist)
    mci = np.eye(12) * 1e3
    tr.bootstrap(s["frames"][0], s["pose_true"][0], s["t"][0], s["ns_true"][0], mci)
    states = []

    def run_step(k):
        j = k % N_FRAMES
        if j == 0:
            r = tr.step(s["frames"][0], s["imu"][0], s["period"], s["pose_true"][0], t_next_last=0.0, reset_ns=s["ns_true"][0], reset_marg=mci)
        else:
            r = tr.step(s["frames"][j], s["imu"][j], s["t"][j], s["pose_true"][j], map_updated=(j == 1 and k > 1))
        states.append(r["state"])
    k = 1
    for _ in range(args.warmup):
        run_step(k); k += 1
    tr.times.clear(); del states[:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(k); k += 1
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    out = dict(units=args.steps, elapsed=elapsed)
    if rank == 0:
        per_call = {name: {"ms_per_call": round(v[0] / v[1] * 1e3, 4), "calls_per_frame": round(v[1] / args.steps, 2)} for name, v in sorted(tr.times.items())}
        in_calls = sum(v[0] for v in tr.times.values())
        out["config"] = {"workload": "ONE EuRoC-shaped synthetic stream %dx%d / %d features through the host-buffer drop-ins call by call (viorb_extract, "
                                     "viorb_preintegrate, viorb_search_by_projection_frame, viorb_pose_opt_vi, viorb_search_by_projection_points, viorb_pose_opt_vi), "
                                     "host buffers in and out of every call; %s; Python caller (numpy glue between the calls = %.2f ms per frame)"
                                     % (cfg["w"], cfg["h"], cfg["nfeat"], "EuRoC lens, keypoints undistorted by viorb_undistort_points" if dist else "pinhole camera",
                                        (elapsed - in_calls) / args.steps * 1e3),
                         "baseline_config": cfg["baseline_config"], "streams_per_gpu": 1, "per_call": per_call,
                         "ms_per_frame_inside_the_calls": round(in_calls / args.steps * 1e3, 4),
                         "frames_per_s_inside_the_calls": round(args.steps / in_calls, 1),
                         "tracked_frames": int(sum(1 for x in states if x == 0)), "frames": len(states)}
        out["roofline"] = None
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline_tracking([s], cfg["w"], cfg["h"], cfg["nfeat"], 100, 10, dist)
    out["metric"] = "frames/sec ORB extract+match+pose-opt, EuRoC 752x480, single stream through the host-buffer drop-ins"
    out["unit"] = "frames/s"; out["dtype"] = "u8"
    return out


# ------------------------------------------------------------------------------------------------------------------------------
# kitti_stereo: extraction of both images + Frame::ComputeStereoMatches (reference src/Frame.cc:241-262, :646-820)
# ------------------------------------------------------------------------------------------------------------------------------
def run_stereo(args, cfg, rank, dev_index, dev, world):
    import ctypes as C
    import torch
    import torch.distributed as dist
    import viorb_amd
    from viorb_amd.capi import lib, check, ptr
    from viorb_amd.synth import make_stereo_pair, KITTI_K
    Pn, W_IMG, H_IMG, NFEAT = args.streams, cfg["w"], cfg["h"], cfg["nfeat"]
    distinct = min(Pn, 8)
    pairs = [make_stereo_pair(100 + rank * 64 + s, W_IMG, H_IMG)[:2] for s in range(distinct)]
    imgs = torch.from_numpy(np.stack([pairs[i % distinct][0] for i in range(Pn)] + [pairs[i % distinct][1] for i in range(Pn)])).to(dev)
    ex = viorb_amd.ORBextractor(NFEAT, 1.2, 8, 20, 7, max_batch=2 * Pn, device=dev_index)
    u = torch.zeros((Pn, ex.cap), dtype=torch.float32, device=dev); d = torch.zeros_like(u); n = torch.zeros(Pn, dtype=torch.int32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def step():
        ex.extract_batch_device(imgs)
        check(lib().viorb_stereo_match_device(ex.h, 0, ex.h, Pn, Pn, KITTI_K["bf"], KITTI_K["fx"], ptr(u), ptr(d), ptr(n), st))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    L = lib()
    L.viorb_profile_select(None if args.all_kernel_events else b"k_fast_cells,k_stereo_match")
    L.viorb_profile_reset(); L.viorb_profile_enable(0 if args.no_kernel_events else 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    L.viorb_profile_enable(0)
    names = C.create_string_buffer(4096); ms = (C.c_double * 64)(); calls = (C.c_int * 64)(); k = C.c_int()
    L.viorb_profile_read(names, 4096, ms, calls, 64, C.byref(k))
    prof = {nm_: (ms[i], calls[i]) for i, nm_ in enumerate(names.value.decode().split("\n")[:k.value])}
    out = dict(units=Pn * args.steps, elapsed=elapsed, metric="stereo pairs/sec: ORB extraction of both images + ComputeStereoMatches, KITTI 1241x376, 2000 features",
               unit="pairs/s", dtype="u8")
    if rank == 0:
        AB, P = algo_bytes(W_IMG, H_IMG, NFEAT)
        roof = None
        if "k_fast_cells" in prof and prof["k_fast_cells"][1]:
            tot_ms, ncalls = prof["k_fast_cells"]; avg_s = tot_ms / ncalls * 1e-3
            per_launch = L.viorb_extractor_fast_launch_images(2 * Pn)    # FAST goes out in sub-launches over the batch; the profiler times one of them per step, in rotation
            bpl = AB["k_fast_cells"] * per_launch
            roof = {"bound": "hbm", "kernel": "k_fast_cells", "achieved": round(bpl / avg_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(bpl / avg_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": None, "avg_launch_us": round(avg_s * 1e6, 2),
                    "algorithmic_bytes_per_launch": int(bpl), "kernel_ms_per_step": {kn: round(v[0] / args.steps, 4) for kn, v in sorted(prof.items())},
                    "launches_per_step": -(-2 * Pn // per_launch), "images_per_launch": per_launch}
            if not args.all_kernel_events:
                roof["kernel_ms_per_step"]["k_fast_cells"] = round(tot_ms / args.steps * (-(-2 * Pn // per_launch)), 4)
        out["roofline"] = roof
        out["config"] = {"workload": "KITTI-shaped synthetic stereo pairs 1241x376, 8 levels, 2000 features per image: extraction of left and right image "
                                     "(one batched handle) + Frame::ComputeStereoMatches (row buckets, Hamming, 11x11 SAD, sub-pixel, median rejection)",
                         "baseline_config": cfg["baseline_config"], "pairs_per_gpu_per_step": Pn, "pairs_per_step": Pn * world,
                         "mean_stereo_matches_per_pair": round(float(n.float().mean().item()), 1), "keypoint_capacity": ex.cap}
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed on rank 0 at N = 1 only
            from oracle import binding as ora
            exo = [ora.Extractor(NFEAT, 1.2, 8, 20, 7), ora.Extractor(NFEAT, 1.2, 8, 20, 7)]
            times = []
            for i in range(3 + 24):
                a, b_ = pairs[i % distinct]
                t1 = time.perf_counter()
                kl, dl = exo[0](a); kr, dr = exo[1](b_)
                ora.stereo_match(exo[0], exo[1], kl, dl, kr, dr, KITTI_K["bf"], KITTI_K["fx"])
                if i >= 3:
                    times.append(time.perf_counter() - t1)
            out["cpu_baseline"] = {"value": round(len(times) / sum(times), 3), "unit": "pairs/s", "cores": 1, "kind": "port",
                                   "sample": "%d pairs after 3 warm-up pairs, oracle (C++ -O3) on 1 host thread (the reference extracts left and right "
                                             "on two threads, src/Frame.cc:258-261): median %.1f ms, mean %.1f ms per pair"
                                             % (len(times), np.median(times) * 1e3, np.mean(times) * 1e3),
                                   "median_ms_per_pair": round(float(np.median(times)) * 1e3, 2), "mean_ms_per_pair": round(float(np.mean(times)) * 1e3, 2),
                                   "hardware_concurrency": os.cpu_count(), "cpu_model": cpu_model()}
    return out


# ------------------------------------------------------------------------------------------------------------------------------
# local_ba: Optimizer::LocalBundleAdjustmentNavState, 20-key-frame window (reference src/Optimizer.cc:1690-2241)
# ------------------------------------------------------------------------------------------------------------------------------
def run_local_ba(args, cfg, rank, dev_index, dev, world):
    import ctypes as C
    import torch
    import torch.distributed as dist
    import viorb_amd
    from viorb_amd.synth import make_local_ba_problem
    from viorb_amd import LocalBundleAdjustmentNavStateBatch
    nwin = args.streams
    # DISTINCT windows (a lock-step group waits for its slowest window; replicas of a few windows all take the same Levenberg path, which is the
    # best case): W in [10, 20] key frames, 1000-3000 points each seen by 3-8 key frames, one seed per window. The IMU blocks are pre-integrated
    # by the library itself (viorb_preintegrate), as the reference's KeyFrame does before LocalMapping calls the solve.
    n_distinct = min(nwin, args.distinct or 64)
    rng = np.random.default_rng(1000 + rank)
    probs = []
    for s in range(n_distinct):
        W = int(rng.integers(10, 21)); npts = int(rng.integers(1000, 3001))
        p = make_local_ba_problem(3 + rank * 4096 + s, W=W, n_points=npts)
        pre = []
        for i, (imu, t0, t1) in enumerate(p["imu"]):
            j = i - 1 if i > 0 else p["prev_kf"]
            pre.append(viorb_amd.preintegrate(imu, p["kfs"][j][10:13], p["kfs"][j][13:16], t0, t1))
        probs.append(dict(kfs=p["kfs"], n_local=p["n_local"], prev_kf=p["prev_kf"], preint=np.stack(pre), points=p["points"], edge_idx=p["edge_idx"],
                          edge_obs=p["edge_obs"], gw=p["gw"], cam=p["cam"]))
    batch = [probs[i % len(probs)] for i in range(nwin)]
    fl = args.in_flight
    for _ in range(max(1, args.warmup)):
        LocalBundleAdjustmentNavStateBatch(batch[:max(fl, 2)], max_in_flight=fl)
    L = viorb_amd.lib()
    L.viorb_profile_select(None); L.viorb_profile_reset(); L.viorb_profile_enable(0 if args.no_kernel_events else 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = LocalBundleAdjustmentNavStateBatch(batch, max_in_flight=fl)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    L.viorb_profile_enable(0)
    out = dict(units=nwin * args.steps, elapsed=elapsed, metric="windows/sec LocalBundleAdjustmentNavState, 10-20-key-frame windows, 1000-3000 points", unit="windows/s", dtype="f64")
    if rank == 0:
        names = C.create_string_buffer(4096); ms = (C.c_double * 64)(); calls = (C.c_int * 64)(); n = C.c_int()
        L.viorb_profile_read(names, 4096, ms, calls, 64, C.byref(n))
        prof = {nm_: (ms[i], calls[i]) for i, nm_ in enumerate(names.value.decode().split("\n")[:n.value]) if nm_}
        # FP64 FLOP model of ONE Levenberg trial of a window: two error passes (60 FLOP per edge each), linearisation with
        # the W blocks and the Hpp / Hll / b sums (450 per edge), IMU + bias factors (10 k per key frame), 3 x 3 inverses (50 per point), the Schur
        # complement gathered per key-frame pair (324 per (point, pair of its observing key frames) entry), the dense Cholesky + two triangular
        # solves of the reduced 12 W system (n^3 / 3 + 2 n^2), back-substitution (36 per edge + 30 per point). Trials >= LM iterations (rejected
        # trials are not reported), so the rate is a lower bound.
        def trial_flop(q):
            ne, npt, Wq = len(q["edge_idx"]), len(q["points"]), int(q["n_local"])
            cnt = np.bincount(np.asarray(q["edge_idx"])[:, 0].astype(np.int64), minlength=npt)      # observations per point (edge_idx rows = (point, key frame))
            pairs = float((cnt * (cnt + 1) / 2).sum())
            nred = 12.0 * Wq
            parts = {"errors": 120.0 * ne, "linearise": 450.0 * ne, "imu": 1.0e4 * Wq, "dinv": 50.0 * npt, "schur": 324.0 * pairs,
                     "cholesky": nred ** 3 / 3 + 2 * nred ** 2, "backsub": 36.0 * ne + 30.0 * npt}
            return parts
        its = np.array([r["its_first"] + r["its_second"] for r in res], np.float64)
        parts = [trial_flop(batch[i]) for i in range(nwin)]
        flop_step = float(sum(sum(pp.values()) * its[i] for i, pp in enumerate(parts)))
        chol_step = float(sum(pp["cholesky"] * its[i] for i, pp in enumerate(parts)))
        ach = flop_step * args.steps / elapsed / 1e12
        kern = {k: round(v[0] / args.steps, 3) for k, v in sorted(prof.items())}
        roof = {"bound": "fp64_vector", "kernel": "viorb_local_ba_navstate_batch (whole lock-step solve)", "achieved": round(ach, 3), "peak": FP64_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / FP64_VECTOR_PEAK_TFLOPS, 5), "traffic": None,
                "flop_per_step": int(flop_step), "kernel_ms_per_step": kern,
                "note": "FP64 FLOP of every LM iteration of every window of a step (model above; rejected trials not counted) over the step's wall time, host "
                        "preparation and transfers included; v_mfma_f64_16x16x4_f64 runs at the FP64 vector rate on gfx950, so one peak serves both. "
                        "MFMA-busy counters: profiles/r04_pmc_mfma_local_ba.txt"}
        if prof.get("k_bab_chol_solve", (0, 0))[1]:
            tms = prof["k_bab_chol_solve"][0]
            roof["cholesky"] = {"kernel": "k_bab_chol_solve", "ms_per_step": round(tms / args.steps, 3), "launches_per_step": prof["k_bab_chol_solve"][1] // args.steps,
                                "achieved": round(chol_step * args.steps / (tms * 1e-3) / 1e12, 3), "unit": "TFLOP/s",
                                "frac": round(chol_step * args.steps / (tms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS, 5),
                                "note": "n^3/3 + 2 n^2 of every iterating window over the kernel's HIP-event time (a launch factors the whole group; the MFMA trailing updates "
                                        "and panel solves are the FLOPs, the diagonal blocks' dependent 1/sqrt chains are the time)"}
        out["roofline"] = roof
        Ws = [int(q["n_local"]) for q in probs]; nps = [len(q["points"]) for q in probs]; nes = [len(q["edge_idx"]) for q in probs]
        out["config"] = {"workload": "synthetic EuRoC-shaped local windows, %d DISTINCT per step (W = %d..%d key frames of 12 unknowns + 4 fixed, %d..%d points seen by 3-8 key "
                                     "frames, %d..%d mono edges, IMU + bias factors); LocalBundleAdjustmentNavState = optimize(5), chi2 gate, optimize(10); "
                                     "viorb_local_ba_navstate_batch, host buffers in and out (its caller is the LocalMapping thread); IMU blocks pre-integrated by "
                                     "viorb_preintegrate" % (n_distinct, min(Ws), max(Ws), min(nps), max(nps), min(nes), max(nes)),
                         "baseline_config": cfg["baseline_config"], "windows_per_gpu_per_step": nwin, "distinct_windows": n_distinct, "windows_in_flight": fl,
                         "lm_iterations_mean_min_max": [round(float(its.mean()), 2), int(its.min()), int(its.max())],
                         "final_chi2_window0": round(float(res[0]["chi2_final"]), 3)}
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed on rank 0 at N = 1 only
            from oracle import binding as ora               # checker, used as the CPU baseline only
            times, ti = [], 0
            t_all = time.perf_counter()
            while time.perf_counter() - t_all < 15.0 and ti < 4 * len(probs):
                q = probs[ti % len(probs)]; a = (q["kfs"], q["n_local"], q["prev_kf"], q["preint"], q["points"], q["edge_idx"], q["edge_obs"], q["gw"], q["cam"])
                t1 = time.perf_counter(); ora.local_ba(*a); times.append(time.perf_counter() - t1); ti += 1
            out["cpu_baseline"] = {"value": round(len(times) / sum(times), 3), "unit": "windows/s", "cores": 1, "kind": "port",
                                   "sample": "%d of the step's distinct windows, one solve each, oracle (C++ -O3, dense Cholesky of the reduced system) on 1 host thread: "
                                             "median %.1f ms" % (len(times), np.median(times) * 1e3),
                                   "median_ms_per_window": round(float(np.median(times)) * 1e3, 2), "hardware_concurrency": os.cpu_count(), "cpu_model": cpu_model()}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 200 for euroc, fewer for the heavier configs)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="euroc", help="BASELINE.json config: euroc = configs[1] (default), kitti_stereo = configs[2], "
                    "local_ba = configs[3], synth720p = configs[4]")
    ap.add_argument("--streams", type=int, default=None, help="independent units per GPU per step: camera streams (euroc 1024, synth720p 8), stereo pairs "
                    "(kitti_stereo 256), windows (local_ba 64)")
    ap.add_argument("--distinct", type=int, default=None, help="tracking configs: distinct synthetic streams generated per rank (default min(streams, 256))")
    ap.add_argument("--gen-procs", type=int, default=None, help="worker processes of the synthetic-stream generator; 1 = in-process, no fork (needed under "
                    "rocprofv3 --pmc, whose preloaded library initialises the GPU before Python starts)")
    ap.add_argument("--in-flight", type=int, default=128, help="local_ba: windows advanced together by the lock-step batch driver (one launch per solver step for the group)")
    ap.add_argument("--no-track-local-map", action="store_true", help="stop after TrackWithIMU's pose solve")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-input", action="store_true", help="tracking configs: frames come from page-locked HOST memory and are uploaded inside the "
                    "timed region (live feed); default: HBM-resident frames, with the live-feed rate reported as config.host_input_frames_per_s")
    ap.add_argument("--no-distortion", action="store_true", help="euroc: pinhole camera instead of the EuRoC lens (the round-1/2 workload)")
    ap.add_argument("--no-host-input-pass", "--no-extra-passes", action="store_true", help="skip the extra passes after the timed region (the other input mode; euroc at N=1: the pinhole-camera rendering)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-kernel HIP events (rooflines become null); dev aid")
    ap.add_argument("--timeline", default=None, help="dev aid (with --all-kernel-events): write start / end / duration (us) of the last kernels to this file")
    ap.add_argument("--timeline-rows", type=int, default=120)
    ap.add_argument("--all-kernel-events", action="store_true", help="time every kernel of the step, not only the roofline kernels (costs ~5 %% of the step)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.streams is None:
        args.streams = cfg["streams"]
    if args.steps is None:
        args.steps = {"euroc": 200, "synth720p": 200, "kitti_stereo": 50, "local_ba": 4, "dropin": 200}[args.config]
    if args.warmup is None:
        args.warmup = {"euroc": 16, "synth720p": 16, "kitti_stereo": 5, "local_ba": 1, "dropin": 16}[args.config]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    from viorb_amd.distributed import reduce_throughput, init as dist_init, stream_seeds
    import torch
    import torch.distributed as dist
    import viorb_amd
    if os.environ.get("VIORB_BENCH_PLUMBING_ONLY"):
        # Everything of the N > 1 path up to (not including) the first GPU call, for the CPU test-suite (tests/test_distributed_cpu.py): argument and
        # rank plumbing, gloo rendezvous, stream ownership, and the {sum units, max time} reduction of a made-up measurement. Prints no metric.
        if world > 1:
            dist_init("gloo")
        full = bool(os.environ.get("VIORB_BENCH_PLUMBING_FULL"))      # the real number of distinct streams per rank instead of 4 (seed ownership at scale)
        n_own = min(args.streams, (args.distinct or 256) if full else 4)
        seeds = stream_seeds(rank, n_own)
        units, elapsed = reduce_throughput(args.streams * args.steps, 1.0 + 0.5 * rank)
        host_pass = args.host_input or (world == 1 and not args.no_host_input_pass)
        mine = {"rank": rank, "local_rank": local_rank, "seeds": seeds if not full else [seeds[0], seeds[-1], len(seeds)], "gen_procs": generator_procs(n_own, args.gen_procs),
                "host_bytes": host_memory_estimate(cfg, args.streams, min(args.streams, args.distinct or 256), world, host_pass) if args.config in ("euroc", "synth720p") else None}
        owned = [None] * world
        if world > 1:
            dist.all_gather_object(owned, mine)
        else:
            owned = [mine]
        if rank == 0:
            print(json.dumps({"plumbing_only": True, "config": args.config, "n_gpus": world, "local_rank": local_rank, "streams_per_gpu": args.streams,
                              "steps": args.steps, "warmup": args.warmup, "units": units, "elapsed": elapsed, "seeds": [o["seeds"] for o in owned],
                              "ranks": owned, "host_cpus": os.cpu_count()}))
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return
    pinhole_pass = args.config == "euroc" and world == 1 and not args.no_distortion and not args.no_host_input_pass
    args.pregenerated = {}
    if args.config in ("euroc", "synth720p"):
        # Every rank's synthetic streams are generated HERE, before the process group (RCCL initialises the device eagerly with device_id=)
        # and before anything else touches the GPU: a process pool forked from a process that has initialised HIP / RCCL is not something to
        # rely on (under rocprofv3 --pmc such workers have been seen not to exit). The pinhole pass (N = 1, after the timed region) too.
        from viorb_amd.synth import EUROC_DIST
        seeds = stream_seeds(rank, min(args.streams, args.distinct or 256))
        lens = EUROC_DIST if (args.config == "euroc" and not args.no_distortion) else None
        args.pregenerated["lens" if lens else "pinhole"] = generate_streams(seeds, cfg["w"], cfg["h"], args.gen_procs, lens)
        if pinhole_pass:
            args.pregenerated["pinhole"] = generate_streams(seeds, cfg["w"], cfg["h"], args.gen_procs, None)
    if viorb_amd.lib().viorb_device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (viorb_amd has no CPU fallback)")
    # VIORB_BENCH_REHEARSAL=1: run the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices, gloo instead of
    # RCCL, which refuses two ranks on one device) — a functional rehearsal, never a measurement
    rehearsal = bool(os.environ.get("VIORB_BENCH_REHEARSAL"))
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        dist_init("gloo" if rehearsal else "nccl", None if rehearsal else dev)       # "nccl" is RCCL on ROCm
    runner = {"euroc": run_tracking, "synth720p": run_tracking, "kitti_stereo": run_stereo, "local_ba": run_local_ba, "dropin": run_dropin}[args.config]
    r = runner(args, cfg, rank, dev_index, dev, world)
    if pinhole_pass:
        # continuity with rounds 1-2, outside the timed region: the same step on the pinhole rendering of the same scenes (those rounds' workload,
        # `--no-distortion`); the lens compresses the periphery, so its frames hold ~30 % more FAST candidates (DESIGN.md "Round 3 measurements")
        import copy
        a2 = copy.copy(args)
        a2.no_distortion = a2.no_cpu_baseline = a2.no_host_input_pass = a2.no_kernel_events = True
        a2.steps = max(8, min(args.steps, 64)); a2.timeline = None
        r2 = run_tracking(a2, cfg, rank, dev_index, dev, 1)
        r["config"]["pinhole_camera_frames_per_s"] = round(r2["units"] / r2["elapsed"], 1)
        r["config"]["pinhole_camera_note"] = "same scenes and step without the lens (--no-distortion, the round-1/2 workload), %d steps after the timed region" % a2.steps
    units, elapsed = reduce_throughput(r["units"], r["elapsed"], None if rehearsal else dev)
    if rank == 0:
        out = {"metric": r["metric"], "value": round(units / elapsed, 2), "unit": r["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": r["dtype"], "data": "synthetic", "config": r.get("config"), "roofline": r.get("roofline")}
        if r.get("roofline_pose") is not None:
            out["roofline_pose"] = r["roofline_pose"]
        out["cpu_baseline"] = r.get("cpu_baseline")
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

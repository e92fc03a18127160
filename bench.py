#!/usr/bin/env python3
"""bench.py — frames/s of the per-frame visual-inertial front-end (ORB extract + SearchByProjection +
IMU pre-integration + PoseOptimization) on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one pass of the hot path over one batch of synthetic frames: one new 752x480 frame for each of
the S independent mono-inertial streams a GPU owns (weak scaling: every rank owns S streams; no data-path
collective, the only RCCL traffic is the frames/time reduction at the end). Inputs (images, IMU samples)
are resident in HBM before the timed region. Rank 0 prints ONE JSON line; `roofline` is measured with HIP
events around the dominant kernel inside the timed region, `cpu_baseline` times the CPU oracle (a port of
the reference's path; the reference itself cannot be built here) on a bounded sample of the same streams.
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

W_IMG, H_IMG, NFEAT, NLEVELS = 752, 480, 1000, 8
N_FRAMES = 8                                  # frames per (periodic) synthetic stream
P_PIXELS = 1117367                            # sum of level pixels, SURVEY.md §8 table (config E)
HBM_PEAK_GBS = 8000.0                         # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ROOFLINE_KERNEL = "k_fast_cells"
# algorithmic bytes per frame of each extractor kernel (B_ext = 4 P + K (709 + 961 + 60))
ALGO_BYTES = {
    "k_fast_cells": P_PIXELS,                                  # 1 P read by FAST
    "k_blur": 2 * P_PIXELS,                                    # 1 P read + 1 P written
    "k_resize": P_PIXELS - W_IMG * H_IMG,                      # levels 1..7 written (reads served from cache)
    "k_copy_level0": W_IMG * H_IMG,                            # level-0 copy
    "k_orient_describe": NFEAT * (709 + 961 + 60),             # patch + window + outputs per keypoint
}


def _gen_stream(seed):
    from viorb_amd.synth import make_periodic_stream
    s = make_periodic_stream(seed, N_FRAMES, W_IMG, H_IMG)
    return dict(frames=s["frames"], imu=s["imu"], t=s["t"], ns_true=s["ns_true"], pose_true=s["pose_true"], period=s["period"],
                cam=s["cam"], gw=s["gw"])


def generate_streams(seeds):
    """CPU-side synthetic data (before anything touches the GPU)."""
    import multiprocessing as mp
    nproc = max(1, min(len(seeds), (os.cpu_count() or 2) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))), 16))
    if nproc == 1:
        return [_gen_stream(s) for s in seeds]
    with mp.get_context("fork").Pool(nproc) as pool:
        return pool.map(_gen_stream, seeds)


def cpu_baseline(streams, budget_s=12.0, max_frames=150, track_local_map=True):
    """The oracle (CPU port of the reference path) on one host core, same streams, same per-frame sequence."""
    from oracle.harness import OracleTracker           # checker only; never on the product path
    done, t_total = 0, 0.0
    for s in streams:
        tr = OracleTracker(s["cam"], s["gw"], W_IMG, H_IMG, NFEAT, track_local_map=track_local_map)
        tr.bootstrap(s["frames"][0], s["pose_true"][0], s["t"][0], s["ns_true"][0], np.eye(12) * 1e3)
        k = 1
        while t_total < budget_s and done < max_frames:
            j = k % N_FRAMES
            t0 = time.perf_counter()
            if j:
                tr.step(s["frames"][j], s["imu"][j], s["t"][j], s["pose_true"][j])
            else:      # the loop closes: key-frame boundary, as in the timed GPU sequence
                tr.step(s["frames"][0], s["imu"][0], s["period"], s["pose_true"][0], t_next_last=0.0, reset_ns=s["ns_true"][0], reset_marg=np.eye(12) * 1e3)
            t_total += time.perf_counter() - t0
            done += 1; k += 1
            if k > 3 * N_FRAMES:
                break
        if t_total >= budget_s or done >= max_frames:
            break
    return done / t_total, done, t_total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 steps = 25 loops of the 8-frame streams, ~0.4 s: the step time is stationary over hundreds of steps (tools/step_times.py)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--streams", type=int, default=256, help="independent camera streams per GPU")
    ap.add_argument("--groups", type=int, default=1, help="split the streams of a GPU into this many independently enqueued groups "
                    "(each with its own HIP streams) so that latency-bound kernels of one group overlap chip-filling kernels of another")
    ap.add_argument("--no-track-local-map", action="store_true", help="stop after TrackWithIMU's pose solve (skip the SearchLocalPoints + "
                    "second PoseOptimization stage of TrackLocalMapWithIMU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-kernel HIP events (roofline becomes null); dev aid "
                    "to measure what the events themselves cost")
    ap.add_argument("--all-kernel-events", action="store_true", help="time every kernel of the step, not only the roofline kernel: fills "
                    "roofline.kernel_ms_per_step for all of them and costs ~5 %% of the step (an event pair is ~8 us of stream time)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    S = args.streams

    # ---- synthetic inputs on the CPU, then the GPU runtime --------------------------------------------
    from viorb_amd.distributed import stream_seeds, reduce_throughput, init as dist_init
    streams = generate_streams(stream_seeds(rank, S))
    import torch
    import torch.distributed as dist
    import viorb_amd
    from viorb_amd.tracker import BatchedTracker
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
    up = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
    frames = up(np.stack([s["frames"] for s in streams], 1))                   # [F, S, h, w] u8
    imu = up(np.stack([s["imu"] for s in streams], 1))                         # [F, S, n, 7] f64
    t_frames = up(np.stack([s["t"] for s in streams], 1))                      # [F, S]
    t_period = up(np.array([s["period"] for s in streams]))                    # [S]
    pose_true = up(np.stack([s["pose_true"] for s in streams], 1))             # [F, S, 12] f64
    ns_true = up(np.stack([s["ns_true"] for s in streams], 1))                 # [F, S, 22]
    zeros_t = torch.zeros(S, dtype=torch.float64, device=dev)
    mci0 = up(np.stack([np.eye(12).ravel() * 1e3] * S))
    cam, gw = streams[0]["cam"], streams[0]["gw"]

    G = max(1, args.groups)
    if S % G:
        raise SystemExit("--streams must be a multiple of --groups")
    Sg = S // G
    sl = [slice(g * Sg, (g + 1) * Sg) for g in range(G)]
    cut = lambda t, g: t[:, sl[g]].contiguous()
    fr_g = [cut(frames, g) for g in range(G)]; imu_g = [cut(imu, g) for g in range(G)]; tf_g = [cut(t_frames, g) for g in range(G)]
    pt_g = [cut(pose_true, g) for g in range(G)]; ns_g = [cut(ns_true, g) for g in range(G)]; tp_g = [t_period[sl[g]].contiguous() for g in range(G)]
    zeros_g = zeros_t[:Sg].contiguous()
    mci_g = [mci0[sl[g]].contiguous() for g in range(G)]
    TLM = not args.no_track_local_map
    trs = [BatchedTracker(cam, gw, Sg, W_IMG, H_IMG, NFEAT, th=15.0, device=dev_index, compute_marg=True, track_local_map=TLM) for _ in range(G)]
    for g, tr in enumerate(trs):
        tr.skip_input_wait = bool(os.environ.get('SKIPWAIT'))
        tr.bootstrap(fr_g[g][0], pt_g[g][0], tf_g[g][0], ns_g[g][0], mci0[sl[g]].contiguous())

    def run_step(k):
        j = k % N_FRAMES
        for g, tr in enumerate(trs):
            if j == 0:      # closing the loop: frame F == frame 0, its stamp is the period; next "last" stamp is 0. It is also the
                # harness's key-frame boundary: the frame is tracked in full, the next one starts from the key frame's state with a
                # fresh prior (an endless frame-to-frame prior chain is not what the reference runs, and it degrades: after ~50 chained
                # frames of this synthetic world the inlier count falls and the LM steps start to be rejected)
                tr.step(fr_g[g][0], imu_g[g][0], tp_g[g], pt_g[g][0], t_next_last=zeros_g, chain_estimate=False, true_ns=ns_g[g][0],
                        marg_reset=mci_g[g])
            else:
                tr.step(fr_g[g][j], imu_g[g][j], tf_g[g][j], pt_g[g][j])

    k = 1
    for _ in range(args.warmup):
        run_step(k); k += 1
    torch.cuda.synchronize()
    L = viorb_amd.lib()
    # the dominant extractor kernel (profiles/*_kernel_stats.csv) is the one the roofline object is about; only its launches carry events
    L.viorb_profile_select(None if args.all_kernel_events else ROOFLINE_KERNEL.encode())
    L.viorb_profile_reset(); L.viorb_profile_enable(0 if args.no_kernel_events else 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(k); k += 1
    t_enq = time.perf_counter() - t0                   # host time to enqueue all steps (the device may still be working)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    L.viorb_profile_enable(0)

    # ---- sanity of the timed work (outside the timed region): every stream tracked its frame ------------
    info = np.concatenate([(tr.info2 if TLM else tr.info).cpu().numpy() for tr in trs])
    n_loc = np.concatenate([tr.n_loc.cpu().numpy() for tr in trs]) if TLM else np.zeros(1)
    nm = np.concatenate([tr.nmatches.cpu().numpy() for tr in trs])
    status_ok = bool(all((tr.status.cpu().numpy() == 0).all() for tr in trs))
    tracked = int((info[:, 0] >= 20).sum())

    # ---- per-kernel HIP-event times ------------------------------------------------------------------------
    import ctypes as C
    names = C.create_string_buffer(4096); ms = (C.c_double * 64)(); calls = (C.c_int * 64)(); n = C.c_int()
    L.viorb_profile_read(names, 4096, ms, calls, 64, C.byref(n))
    prof = {nm_: (ms[i], calls[i]) for i, nm_ in enumerate(names.value.decode().split("\n")[:n.value])}
    ext = {kname: v for kname, v in prof.items() if kname in ALGO_BYTES}
    dom = ROOFLINE_KERNEL if ROOFLINE_KERNEL in ext else None

    # ---- reduce over ranks: total frames, max time ----------------------------------------------------------
    frames_done, elapsed = reduce_throughput(S * args.steps, elapsed, None if rehearsal else dev)

    if rank == 0:
        roof = None
        if dom:
            tot_ms, ncalls = ext[dom]
            launches_per_step = ncalls / args.steps
            avg_s = tot_ms / ncalls * 1e-3
            bytes_per_launch = ALGO_BYTES[dom] * S / launches_per_step
            achieved = bytes_per_launch / avg_s / 1e9
            ext_ms = sum(v[0] for v in ext.values())
            ext_bytes = (4 * P_PIXELS + NFEAT * (709 + 961 + 60)) * S * args.steps
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "avg_launch_us": round(avg_s * 1e6, 2), "algorithmic_bytes_per_launch": int(bytes_per_launch),
                    "extractor_all_kernels_GBps": round(ext_bytes / (ext_ms * 1e-3) / 1e9, 2) if args.all_kernel_events else None,
                    "kernel_ms_per_step": {kname: round(v[0] / args.steps, 4) for kname, v in sorted(prof.items())}}
        cpu = None
        if not args.no_cpu_baseline:
            fps, nfr, tsec = cpu_baseline(streams[:8], track_local_map=TLM)
            cpu = {"value": round(fps, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": "%d frames of the same synthetic streams, same per-frame sequence, oracle (C++ -O3) on 1 host thread, %.1f s"
                             % (nfr, tsec)}
        out = {
            "metric": "frames/sec ORB extract+match+pose-opt, EuRoC 752x480, 1/2/4/8 GPUs",
            "value": round(frames_done / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "EuRoC-shaped synthetic mono-inertial streams 752x480, 8 levels, 1000 features, per frame: extract + IMU "
                                   "pre-integration (10 samples) + SearchByProjection(th=15) + PoseOptimization(Frame,Frame)" +
                                   (" + SearchLocalPoints(~2000 local points, th=1) + PoseOptimization(Frame,Frame,marg)  [TrackWithIMU + TrackLocalMapWithIMU]"
                                    if TLM else " with marginal  [TrackWithIMU only]"),
                       "keyframe_boundary_every_frames": N_FRAMES,
                       "track_local_map": TLM, "mean_local_matches_last_step": round(float(n_loc.mean()), 1),
                       "streams_per_gpu": S, "stream_groups_per_gpu": G, "frames_per_step": S * world, "solver_dtype": "f64",
                       "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3), "tracked_streams_last_step": tracked, "mean_matches_last_step": round(float(nm.mean()), 1),
                       "mean_inliers_last_step": round(float(info[:, 0].mean()), 1), "status_ok": status_ok},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""N > 1 path on CPU: world_size-2 gloo run of the sharding + throughput reduction bench.py uses on GPUs."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from viorb_amd.distributed import stream_seeds, reduce_throughput, init
    d = init("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    seeds = stream_seeds(rank, 4)
    gathered = [None] * world
    dist.all_gather_object(gathered, seeds)
    frames, elapsed = reduce_throughput(4 * 10, 1.0 + 0.5 * rank)        # rank 1 is the slow one
    if rank == 0:
        print(json.dumps({"seeds": gathered, "frames": frames, "elapsed": elapsed, "world": world}))
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gloo_sharding_and_reduction(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2
    flat = [s for part in r["seeds"] for s in part]
    assert flat == list(range(1000, 1008))                      # disjoint, gap-free stream ownership
    assert r["frames"] == 80.0 and abs(r["elapsed"] - 1.5) < 1e-12     # sum of frames, max of time


def test_single_process_reduction_is_identity():
    sys.path.insert(0, ROOT)
    from viorb_amd.distributed import reduce_throughput, stream_seeds
    assert reduce_throughput(640, 0.25) == (640.0, 0.25)
    assert stream_seeds(3, 2) == [1006, 1007]


def test_bench_plumbing_on_two_gloo_ranks():
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one rank per GPU, --gpus 2 --steps K --warmup W), with
    VIORB_BENCH_PLUMBING_ONLY=1 stopping it before the first GPU call: the ranks need nothing but RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*,
    own disjoint streams, and rank 0 prints one JSON object with the summed units and the slowest rank's time."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", VIORB_BENCH_PLUMBING_ONLY="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0 only
    r = json.loads(lines[0])
    assert r["plumbing_only"] and r["n_gpus"] == 2 and r["local_rank"] == 0 and r["steps"] == 5 and r["warmup"] == 2
    assert r["units"] == 2 * r["streams_per_gpu"] * 5 and abs(r["elapsed"] - 1.5) < 1e-12
    assert [x for part in r["seeds"] for x in part] == list(range(1000, 1008))
    # WORLD_SIZE and --gpus must agree
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert bad.returncode != 0


def test_bench_plumbing_on_eight_gloo_ranks():
    """The driver's N = 8 launch of bench.py on CPU (gloo, VIORB_BENCH_PLUMBING_ONLY): eight ranks with the REAL number of distinct streams per rank
    own disjoint, gap-free seed ranges; local ranks are 0..7; the synthetic-stream generators of the eight ranks together do not oversubscribe the
    host (LOCAL_WORLD_SIZE-aware process counts); a rank's host memory stays bounded (no page-locked copy of the frames at N > 1)."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", VIORB_BENCH_PLUMBING_ONLY="1", VIORB_BENCH_PLUMBING_FULL="1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["units"] == 8 * r["streams_per_gpu"] * 3 and abs(r["elapsed"] - (1.0 + 0.5 * 7)) < 1e-12     # sum of units, slowest rank's time
    ranks = sorted(r["ranks"], key=lambda q: q["rank"])
    assert [q["rank"] for q in ranks] == list(range(8)) and [q["local_rank"] for q in ranks] == list(range(8))
    nxt = 1000
    for q in ranks:                                               # [first, last, count] per rank: consecutive blocks of 256 seeds
        first, last, cnt = q["seeds"]
        assert first == nxt and cnt == 256 and last == first + cnt - 1
        nxt = last + 1
    cpus = r["host_cpus"]
    assert all(q["gen_procs"] == max(1, min(256, cpus // 8, 16)) for q in ranks)
    assert sum(q["gen_procs"] for q in ranks) <= max(cpus, 8)
    for q in ranks:
        hb = q["host_bytes"]
        assert hb["page_locked_copy"] == 0
        assert hb["generated"] + hb["upload_stack"] < 2 * 2 ** 30        # 256 distinct streams + their upload stack (replicated to 1024 on the device): 1.5 GB

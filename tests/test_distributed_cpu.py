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

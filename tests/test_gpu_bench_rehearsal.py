"""bench.py's N > 1 code path end to end on the one GPU of the test box: two ranks share the device (VIORB_BENCH_REHEARSAL=1: gloo instead of
RCCL, which refuses two ranks on one device). A functional rehearsal of what the driver launches on the 8-GPU node — rank plumbing, disjoint
stream seeds, barrier-bracketed timing, {sum units, max time} reduction, one JSON line from rank 0 — never a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("config,extra", [("euroc", ["--streams", "16", "--distinct", "4"]), ("kitti_stereo", ["--streams", "4"])])
def test_bench_two_ranks_on_one_gpu(config, extra):
    env = dict(os.environ, VIORB_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", config, "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-extra-passes",
           "--gen-procs", "1"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    per_rank = int(extra[1])
    assert abs(d["value"] - 2 * per_rank * 4 / (d["ms_per_step"] * 4e-3)) < 0.01 * d["value"]     # units of BOTH ranks over the slowest rank's time
    if config == "euroc":
        assert d["config"]["frames_per_step"] == 2 * per_rank and d["config"]["status_ok"] and d["config"]["tracked_streams_last_step"] == per_rank

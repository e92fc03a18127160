"""-m gpu: the RCCL half of the multi-GPU layer on the one GPU of the test box. The driver launches `bench.py` with one rank per GPU over
backend "nccl" (= RCCL on ROCm); with a single GPU only a world of ONE rank can use that backend (RCCL refuses two ranks on one device), which
still covers what the gloo tests cannot: process-group creation with `device_id=` (viorb_amd/distributed.py: init), the f64 MAX / SUM
all-reduces of the throughput reduction on device tensors, and the barrier that brackets bench.py's timed region."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from viorb_amd.distributed import init
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    init("nccl", dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1 and dist.get_rank() == 0
    t = torch.tensor([3.25], dtype=torch.float64, device=dev); f = torch.tensor([51200.0], dtype=torch.float64, device=dev)
    dist.barrier(); torch.cuda.synchronize()
    dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.barrier(); torch.cuda.synchronize()
    assert t.item() == 3.25 and f.item() == 51200.0
    dist.destroy_process_group()
    print("rccl ok")
""") % ROOT


@pytest.mark.gpu
def test_rccl_process_group_of_one_rank():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])

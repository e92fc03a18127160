"""Multi-GPU layer of the batched many-stream mode. Streams share nothing, so the path shards by
stream: rank r owns streams [r*S, (r+1)*S) (weak scaling) and there is no data-path collective. The only
collective is the throughput reduction {frames: sum, elapsed: max} after the timed region — RCCL over xGMI on
GPUs (backend "nccl"), gloo in the CPU tests."""
import os


def stream_seeds(rank, streams_per_rank, base=1000):
    """Disjoint, gap-free seed ranges: the global stream id is the seed offset."""
    return [base + rank * streams_per_rank + i for i in range(streams_per_rank)]


def init(backend, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    return dist


def reduce_throughput(frames_done, elapsed_s, device=None):
    """Whole-job (frames, elapsed): frames summed over ranks, elapsed = max over ranks. No-op without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(frames_done), float(elapsed_s)
    kw = {"device": device} if device is not None else {}
    tt = torch.tensor([elapsed_s], dtype=torch.float64, **kw)
    ff = torch.tensor([float(frames_done)], dtype=torch.float64, **kw)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dist.all_reduce(ff, op=dist.ReduceOp.SUM)
    return ff.item(), tt.item()

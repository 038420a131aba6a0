"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm,
"gloo" on CPU for tests).  Inference and voting shard by IMAGE with no exchange on the data
path: ranks are independent replicas; the only collectives are the benchmark's
barrier and its max-over-ranks clock.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def force_collectives() -> bool:
    """CASAPOSE_DIST_FORCE=1: create the process group and run every collective of the data-parallel protocol even with ONE rank, so that
    the RCCL code path (group creation, barrier with device ids, fp64 statistic all-reduces, asynchronous gradient buckets on RCCL's
    stream) executes on a single-GPU box.  A SUM over one rank is the identity: results must equal the plain single-process run."""
    return os.environ.get("CASAPOSE_DIST_FORCE", "0") == "1"


def init_from_env(backend: str = "nccl") -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE (torchrun contract); returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # CASAPOSE_DIST_BACKEND=gloo lets several ranks share ONE GPU in tests (RCCL refuses duplicate devices)
    backend = os.environ.get("CASAPOSE_DIST_BACKEND", backend)
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of `total` images for `rank`; sizes differ by at most one and
    the union over ranks is exactly range(total) (ragged tails included)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world: %d/%d" % (rank, world))
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def barrier_sync(device=None) -> None:
    if dist.is_initialized():
        if dist.get_backend() == "nccl" and device is not None and torch.device(device).type == "cuda":
            dist.barrier(device_ids=[torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()])
        else:
            dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_reduce_sum_(t: torch.Tensor, group=None, world_size: int = 1) -> torch.Tensor:
    """In-place SUM all-reduce (RCCL on GPU tensors, gloo on CPU tensors); no-op for a single replica.  The training
    step uses it for (a) the fp64 SyncBN statistic tables -- forward sums and the two backward means -- and (b) the
    flat fp32 gradient before Adam (MirroredStrategy's SUM reduction, train_casapose.py:641-643)."""
    if world_size > 1 or (force_collectives() and dist.is_initialized()):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_sum_async(t: torch.Tensor, group=None):
    """Start a SUM all-reduce of `t` and return the work handle (`.wait()` makes the current stream wait for it).  RCCL runs it on its
    own stream after the work already queued on the current stream, so it overlaps whatever is launched next."""
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


def reduce_step_log(losses, pose_stats, world_size: int, device=None):
    """The per-step logging exchange of the training driver as ONE collective: `losses` (5 python floats) become their MEAN over the
    replicas (strategy.reduce(MEAN), train_casapose.py:690-694), `pose_stats` (a sequence of >= 6 per-object vectors, or None) its first
    six vectors SUMMED over the replicas (:732-737).  Returns (list of floats, [6, objects] float64 array or None)."""
    import numpy as np

    st = None if pose_stats is None else np.stack([np.asarray(pose_stats[j], np.float64).reshape(-1) for j in range(6)])
    if world_size <= 1 or not dist.is_initialized():
        return [float(v) for v in losses], st
    flat = np.concatenate([np.asarray(losses, np.float64) / world_size] + ([st.reshape(-1)] if st is not None else []))
    t = torch.from_numpy(flat).to(device if device is not None and dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    flat = t.cpu().numpy()
    n = len(losses)
    return [float(v) for v in flat[:n]], (flat[n:].reshape(st.shape) if st is not None else None)

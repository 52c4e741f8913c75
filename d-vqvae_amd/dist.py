"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" is RCCL on ROCm, xGMI
between the 8 GPUs of a node).  The path shards over the batch-of-objects dimension -- objects are fully
independent -- so the only exchange is ONE all-gather of the generated MANO parameters
``[B/R, 61]`` (2.0 MB per rank at B=65536, R=8: latency-bound on the fully connected xGMI mesh)."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the launcher's environment (torchrun), single-process defaults."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: Optional[str] = None, share_gpu: bool = False) -> Tuple[int, int, int]:
    """``share_gpu``: every rank drives cuda:0 (plumbing tests of the N > 1 path on a one-GPU box; needs "gloo",
    RCCL refuses two ranks on one device).  Returns (rank, local_rank or 0, world)."""
    rank, local_rank, world = env_world()
    if share_gpu:
        local_rank = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() and not share_gpu else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row range [lo, hi) of rank ``rank``: the first ``total % world`` ranks get one extra row."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, total_rows: Optional[int] = None) -> torch.Tensor:
    """Rank-major concatenation of every rank's ``[rows_r, C]`` tensor (ragged shards allowed)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    local = local.contiguous()
    if local.is_cuda and dist.get_backend() == "gloo":       # gloo gathers host tensors only (one-GPU plumbing tests)
        return all_gather_rows(local.cpu(), total_rows).to(local.device)
    if total_rows is not None and total_rows % world == 0 and local.shape[0] * world == total_rows:
        out = torch.empty((total_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out
    if total_rows is None:                                   # shard sizes unknown: exchange them first (one tiny all-gather)
        counts = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
        counts = [int(v) for v in counts.tolist()]
        sizes, lo = [], 0
        for n in counts:
            sizes.append((lo, lo + n))
            lo += n
    else:
        sizes = [shard_range(total_rows, r, world) for r in range(world)]
        if sizes[rank][1] - sizes[rank][0] != local.shape[0]:
            raise RuntimeError(f"all_gather_rows: rank {rank} holds {local.shape[0]} rows, shard_range says {sizes[rank][1] - sizes[rank][0]}")
    pad = max(hi - lo for lo, hi in sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = torch.empty((pad * world,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    return torch.cat([out[r * pad: r * pad + (hi - lo)] for r, (lo, hi) in enumerate(sizes)])


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

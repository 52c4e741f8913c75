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


_force_collectives = False     # world size 1 with a live process group: run the collectives anyway (first-contact tests of RCCL)


def init(backend: Optional[str] = None, share_gpu: bool = False, force_group: bool = False) -> Tuple[int, int, int]:
    """``share_gpu`` (or DVQ_SHARE_GPU=1): every rank drives cuda:0 (plumbing tests of the N > 1 path on a one-GPU box;
    needs "gloo", RCCL refuses two ranks on one device).  ``force_group`` (or DVQ_FORCE_PG=1): create the process group and
    run every collective even at world size 1, so that RCCL initialisation, all_gather_into_tensor, barrier and
    all_reduce execute on a one-GPU box.  ``backend`` defaults to DVQ_DIST_BACKEND, then "nccl" (= RCCL) on GPUs.
    Returns (rank, local_rank or 0, world)."""
    global _force_collectives
    rank, local_rank, world = env_world()
    share_gpu = share_gpu or os.environ.get("DVQ_SHARE_GPU") == "1"
    force_group = force_group or os.environ.get("DVQ_FORCE_PG") == "1"
    backend = backend or os.environ.get("DVQ_DIST_BACKEND") or None
    if share_gpu:
        local_rank = 0
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() and not share_gpu else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    _force_collectives = bool(force_group) and dist.is_initialized()
    return rank, local_rank, world


def _active() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or _force_collectives)


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row range [lo, hi) of rank ``rank``: the first ``total % world`` ranks get one extra row."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


_comms = {}          # device index -> communicator handle of the C ABI (dvq_comm_init)
collective_used = "none (single process)"       # what the last data collective went through (bench.py reports it)


def _abi_comm(device: torch.device):
    """This process' RCCL communicator behind the C ABI (include/dvq.h: dvq_comm_*), created at first use: rank 0 makes the
    unique id, the process group carries it to the other ranks."""
    import ctypes as C
    from . import _lib
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _comms:
        return _comms[idx]
    lib = _lib.load()
    world, rank = dist.get_world_size(), dist.get_rank()
    # Every rank must end up with the same answer: a rank whose librccl cannot be resolved (or whose init fails) must not raise while
    # the others sit in ncclCommInitRank or in the first collective.  So: every rank probes RCCL (dvq_comm_available resolves the
    # symbols; rank 0 alone makes THE id), the communicator is only initialised if EVERY rank could (all-reduce over the process group) and only
    # used if EVERY rank's init succeeded (a second one); otherwise all ranks fall back to torch.distributed's all-gather together
    # (the None is cached: asked once per device).
    uid = C.create_string_buffer(128)
    have = lib.dvq_comm_available() == 1                   # resolves the symbols, creates nothing
    if have and rank == 0:                                 # only rank 0 makes an id (RCCL starts its bootstrap root for it)
        have = lib.dvq_comm_unique_id(uid, 128) == 0
    box = [bytes(uid.raw) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)

    def everybody(flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    handle, good = C.c_void_p(), False
    if everybody(have):
        # ncclCommInitRank blocks until every rank has joined the bootstrap: a rank that never arrives (a crashed peer, a blocked
        # socket) would leave this one hanging for ever.  The call runs on a helper thread; if it has not returned after
        # DVQ_COMM_TIMEOUT seconds (default 180) this process says so and EXITS non-zero -- the launcher then tears the job down.
        import threading
        result = {}

        def init():
            with torch.cuda.device(idx):
                result["rc"] = lib.dvq_comm_init(box[0], 128, world, rank, C.byref(handle))
        th = threading.Thread(target=init, daemon=True)
        th.start()
        th.join(float(os.environ.get("DVQ_COMM_TIMEOUT", "180")))
        if th.is_alive():
            import sys
            print(f"[dvq dist] rank {rank}/{world}: dvq_comm_init (RCCL bootstrap) did not return within the time limit; exiting",
                  file=sys.stderr, flush=True)
            os._exit(70)
        good = result.get("rc", 1) == 0
        if not everybody(good) and good:
            lib.dvq_comm_destroy(handle)
            good = False
    _comms[idx] = handle if good else None
    return _comms[idx]


def comm_ranks_seen(device: Optional[torch.device] = None) -> Optional[int]:
    """Ranks of the C ABI's RCCL communicator as RCCL itself reports them (ncclCommCount through dvq_comm_count), or None when the
    last collectives went through torch.distributed (gloo, the fallback, one process)."""
    import ctypes as C
    from . import _lib
    if not _comms:
        return None
    idx = (device.index if device is not None and device.index is not None else torch.cuda.current_device())
    h = _comms.get(idx)
    if h is None:
        return None
    n = C.c_int(-1)
    _lib.check(_lib.load().dvq_comm_count(h, C.byref(n)), "dvq_comm_count")
    return int(n.value)


def _gather_equal(local: torch.Tensor, world: int) -> torch.Tensor:
    """one all-gather of equal [rows, C] shards straight into the result: through the C ABI's dvq_allgather_params (RCCL on the
    caller's stream) for float32 device tensors under the "nccl" backend, through torch.distributed otherwise"""
    global collective_used
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    use_abi = (local.is_cuda and local.dtype == torch.float32 and local.dim() == 2 and dist.get_backend() == "nccl"
               and os.environ.get("DVQ_ALLGATHER", "abi") != "torch")
    comm = _abi_comm(local.device) if use_abi else None      # None: some rank could not build the communicator (all ranks agree)
    if comm is not None:
        from . import _lib
        with torch.cuda.device(local.device):
            _lib.check(_lib.load().dvq_allgather_params(comm, local.data_ptr(), local.shape[0], local.shape[1], out.data_ptr(),
                                                        torch.cuda.current_stream(local.device).cuda_stream), "dvq_allgather_params")
        collective_used = "dvq_allgather_params (C ABI -> RCCL ncclAllGather on the caller's stream)"
    else:
        dist.all_gather_into_tensor(out, local)
        collective_used = f"torch.distributed {dist.get_backend()} all_gather_into_tensor"
    return out


def all_gather_rows(local: torch.Tensor, total_rows: Optional[int] = None, verify: bool = True) -> torch.Tensor:
    """Rank-major concatenation of every rank's ``[rows_r, C]`` tensor (ragged shards allowed).

    Every rank takes the same sequence of collectives whatever its own row count: the shard sizes are exchanged first
    (one all-gather of ``world`` int64) and the data path -- one equal-size all-gather, or a padded one -- follows from the
    exchanged sizes alone, so a rank whose rows disagree with ``total_rows`` makes EVERY rank raise instead of leaving the
    others inside a collective.  ``verify=False`` with ``total_rows``: the caller guarantees rank r holds
    ``shard_range(total_rows, r, world)`` rows (checked locally) and the exchange + its host sync are skipped -- the timed
    loop of bench.py after one verified call."""
    if not _active():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    local = local.contiguous()
    if local.is_cuda and dist.get_backend() == "gloo":       # gloo gathers host tensors only (one-GPU plumbing tests)
        return all_gather_rows(local.cpu(), total_rows, verify).to(local.device)
    if total_rows is not None and not verify:
        sizes = [shard_range(total_rows, r, world) for r in range(world)]
        if sizes[rank][1] - sizes[rank][0] != local.shape[0]:
            raise RuntimeError(f"all_gather_rows(verify=False): rank {rank} holds {local.shape[0]} rows, shard_range says "
                               f"{sizes[rank][1] - sizes[rank][0]}")
    else:
        counts = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
        counts = [int(v) for v in counts.tolist()]
        if total_rows is not None and sum(counts) != total_rows:       # the same verdict on every rank
            raise RuntimeError(f"all_gather_rows: the ranks hold {counts} rows, {sum(counts)} in all, expected {total_rows}")
        sizes, lo = [], 0
        for n in counts:
            sizes.append((lo, lo + n))
            lo += n
    total = sizes[-1][1]
    pad = max(hi - lo for lo, hi in sizes)
    if all(hi - lo == pad for lo, hi in sizes):               # equal shards: one collective straight into the result
        return _gather_equal(local, world)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = _gather_equal(buf, world)                           # ragged: padded shards, the padding cut out afterwards
    return torch.cat([out[r * pad: r * pad + (hi - lo)] for r, (lo, hi) in enumerate(sizes)])


def gather_objects(obj):
    """every rank's small Python object, rank-major (one process: [obj])"""
    if not _active():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def barrier():
    if _active():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not _active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    if _comms:
        from . import _lib
        for h in _comms.values():
            if h is not None:
                _lib.load().dvq_comm_destroy(h)
        _comms.clear()
    if dist.is_initialized():
        dist.destroy_process_group()

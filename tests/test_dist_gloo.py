"""Multi-process path on CPU: world_size 2 over gloo (the GPU job uses the same code over RCCL)."""
import os
import socket

import torch
import torch.distributed as td
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from dvqvae_amd import dist
    r, lr, w = dist.init(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = dist.shard_range(total, rank, world)
    full = torch.arange(total * 61, dtype=torch.float32).view(total, 61)
    gathered = dist.all_gather_rows(full[lo:hi].clone(), total_rows=total)
    ok = torch.equal(gathered, full)
    mx = dist.max_over_ranks(float(rank + 1), "cpu")
    dist.barrier()
    ret[rank] = (ok, mx)
    td.destroy_process_group()


def _run(total):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, total, ret), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world)), "all-gathered parameters differ from the unsharded tensor"
    assert all(ret[r][1] == 2.0 for r in range(world))


def test_all_gather_even_shards():
    _run(64)


def test_all_gather_ragged_shards():
    _run(37)

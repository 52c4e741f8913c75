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


def _params(total):
    """[total, 61] MANO parameter rows as the path produces them: the reference's own outputs for 8 objects (fixture G7:
    recon [8,55], recon_pos [8,6]) through the oracle's 61-parameter assembly, repeated with a per-row offset."""
    import numpy as np
    from oracle import dvq_oracle as O
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_gen.npz"))
    p8 = O.assemble61(torch.from_numpy(g["recon"]), torch.from_numpy(g["recon_pos"]))
    reps = (total + 7) // 8
    return (p8.repeat(reps, 1) + torch.arange(reps * 8, dtype=torch.float32)[:, None] * 1e-3)[:total].contiguous()


def _worker(rank, world, port, total, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from dvqvae_amd import dist
    r, lr, w = dist.init(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = dist.shard_range(total, rank, world)
    full = _params(total)
    gathered = dist.all_gather_rows(full[lo:hi].clone(), total_rows=total)
    ok = torch.equal(gathered, full)
    ok = ok and torch.equal(dist.all_gather_rows(full[lo:hi].clone()), full)         # shard sizes exchanged first
    uneven = full[: total // 3] if rank == 0 else full[total // 3:]                 # shards shard_range would not produce
    ok = ok and torch.equal(dist.all_gather_rows(uneven.clone()), full)
    mx = dist.max_over_ranks(float(rank + 1), "cpu")
    rows = dist.gather_objects([lo, hi])                                            # the bench line's rows_per_rank
    ok = ok and rows == [list(dist.shard_range(total, q, world)) for q in range(world)]
    ok = ok and dist.comm_ranks_seen() is None                                      # no RCCL communicator under gloo
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


def test_shard_ranges_cover_the_batch_in_rank_major_order():
    from dvqvae_amd import dist
    for total, world in ((65536, 8), (65536, 4), (37, 8), (5, 8), (0, 2)):
        spans = [dist.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def _mismatch_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dvqvae_amd import dist
    dist.init(backend="gloo")
    full = _params(10)
    mine = full[:5] if rank == 0 else full[5:9]                       # rank 1 is one row short of total_rows = 10
    try:
        dist.all_gather_rows(mine.clone(), total_rows=10)
        ret[rank] = "no error"
    except RuntimeError as e:
        ret[rank] = str(e)
    dist.barrier()                                                    # both ranks are still in step: nobody hangs in a collective
    td.destroy_process_group()


def test_row_count_mismatch_raises_on_every_rank_instead_of_hanging():
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_mismatch_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all("expected 10" in ret[r] for r in range(world)), dict(ret)


def _forced_worker(rank, world, port, ret):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dvqvae_amd import dist
    r, lr, w = dist.init(backend="gloo", force_group=True)            # world size 1, collectives executed anyway
    full = _params(12)
    ok = td.is_initialized() and w == 1 and torch.equal(dist.all_gather_rows(full.clone(), total_rows=12), full)
    ok = ok and torch.equal(dist.all_gather_rows(full.clone(), total_rows=12, verify=False), full)
    ok = ok and dist.max_over_ranks(3.5, "cpu") == 3.5
    dist.barrier()
    dist.shutdown()
    ret[0] = ok and not td.is_initialized()


def test_forced_process_group_at_world_size_one():
    ret = mp.Manager().dict()
    mp.spawn(_forced_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert ret[0]

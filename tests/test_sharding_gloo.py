"""World-size-2 gloo rehearsal (CPU) of bench.py's multi-GPU scheme: tree samples shard over ranks with
no data-path collective, and ONE gather of the per-sample log-likelihoods reaches rank 0 in rank order.
The evaluation itself is replaced by a deterministic stand-in (the HIP path needs a GPU); what is under
test is the sharding, the gather layout and the max-over-ranks timing reduction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(n_total, world, rank):
    """Contiguous weak-scaling shards: rank r owns samples [r*n, (r+1)*n)."""
    n = n_total // world
    return rank * n, n


def _fake_loglik(sample_ids):
    return -1000.0 - 0.25 * sample_ids.to(torch.float64)


def _worker(rank, world, port, n_per_rank, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, n = _shard(world * n_per_rank, world, rank)
    ids = torch.arange(start, start + n)
    ll = _fake_loglik(ids)
    gathered = [torch.zeros(n, dtype=torch.float64) for _ in range(world)] if rank == 0 else None
    dist.gather(ll, gather_list=gathered, dst=0)
    t = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)   # pretend rank 1 is the slowest
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(out_path, np.concatenate([g.numpy() for g in gathered] + [t.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    world, n = 2, 37
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    got = np.load(out)
    want = _fake_loglik(torch.arange(world * n)).numpy()
    np.testing.assert_array_equal(got[:-1], want)      # rank order == sample order, nothing lost
    assert got[-1] == pytest.approx(0.020)             # MAX over ranks

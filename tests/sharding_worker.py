"""Rank process of tests/test_sharding_gloo.py: runs bench.py's multi-rank flow (linearham_amd.sharding:
take_shard -> evaluate -> gather_loglik -> unshard, max_over_ranks) over gloo on CPU, with a deterministic
stand-in for the HIP evaluation (which needs a GPU).  Started by sharding.spawn_ranks; rank 0 prints JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from linearham_amd import sharding  # noqa: E402


def table(n_rows):
    """A stand-in for the flattened RevBayes table: per-row arrays keyed like host.PhyloHMM.flatten_tsv."""
    rng = np.random.default_rng(5)
    return {"n_rows": n_rows, "ops": rng.integers(0, 9, size=(n_rows, 3, 4)).astype(np.int32),
            "brlen": rng.random((n_rows, 8)), "er": rng.random((n_rows, 6)), "pi": rng.random((n_rows, 4)),
            "alpha": rng.random(n_rows)}


def fake_loglik(d):
    return -1000.0 - d["brlen"].sum(1) - 3.0 * d["er"][:, 2] - d["alpha"] - d["ops"].reshape(len(d["alpha"]), -1).sum(1)


def main():
    n_rows, n_total = int(sys.argv[1]), int(sys.argv[2])
    fail_rank = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    rank, local_rank, world = sharding.rank_env()
    if rank == fail_rank:
        sys.exit(3)
    dist.init_process_group("gloo")
    flat = table(n_rows)
    shard, ids = sharding.take_shard(flat, n_total, world, rank)
    ll = torch.from_numpy(fake_loglik(shard))
    res = sharding.gather_loglik(ll, n_total, world, rank, "gloo")
    t = sharding.max_over_ranks(0.010 * (rank + 1), world, "gloo")
    if rank == 0:
        got = sharding.unshard(res.numpy(), n_total, world)
        print(json.dumps({"world": world, "loglik": got.tolist(), "t_max": t, "n_rank0": len(ids)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

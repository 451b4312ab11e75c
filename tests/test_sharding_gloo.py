"""World-size-2/3 gloo runs (CPU) of the product's multi-GPU scheme, linearham_amd/sharding.py -- the
module bench.py uses: sample i -> rank i mod N, no data-path collective, ONE gather of the per-sample
log-likelihoods to rank 0, MAX-over-ranks timing, and the launcher that starts the rank processes.  Only
the evaluation is a stand-in (the HIP path needs a GPU)."""
import json
import os
import sys

import numpy as np
import pytest

from linearham_amd import sharding

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "sharding_worker.py")


def test_shard_ids_partition_every_sample_once():
    for n_total, world in ((10, 1), (10, 3), (7, 8), (24576 * 8, 8), (10000, 8)):
        seen = np.concatenate([sharding.shard_ids(n_total, world, r) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(n_total))
        for r in range(world):
            ids = sharding.shard_ids(n_total, world, r)
            assert len(ids) == sharding.shard_size(n_total, world, r)
            assert all(int(i) % world == r for i in ids)          # sample i -> GPU i mod N
    with pytest.raises(ValueError):
        sharding.shard_ids(4, 2, 2)


def test_table_reuse_keeps_every_rank_on_distinct_rows():
    n_rows, world, batch = 24576, 8, 24576
    for r in range(world):
        ids = sharding.shard_ids(world * batch, world, r)
        rows = sharding.table_rows(ids, n_rows, world, world * batch)
        assert len(set(rows.tolist())) == n_rows                  # a whole pass over the table, not 8 x an eighth
        # the mapping of one sample does not depend on which other samples are asked about
        assert sharding.table_rows(ids[:3], n_rows, world, world * batch).tolist() == rows[:3].tolist()
    # no reuse: sample g reads row g
    np.testing.assert_array_equal(sharding.table_rows(np.arange(100), 10000, 8, 10000), np.arange(100))


def test_unshard_inverts_sharding_with_padding():
    n_total, world = 11, 4
    vals = np.arange(n_total) * 1.5
    m = sharding.shard_size(n_total, world, 0)
    parts = []
    for r in range(world):
        p = np.full(m, np.nan)
        ids = sharding.shard_ids(n_total, world, r)
        p[:len(ids)] = vals[ids]
        parts.append(p)
    np.testing.assert_array_equal(sharding.unshard(parts, n_total, world), vals)


@pytest.mark.parametrize("world,n_rows,n_total", [(2, 37, 37), (2, 16, 41), (3, 10, 10)])
def test_ranks_shard_gather_in_sample_order(world, n_rows, n_total):
    status, out = sharding.spawn_ranks([sys.executable, WORKER, str(n_rows), str(n_total)], world, timeout_s=240)
    assert status == 0, out
    got = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    sys.path.insert(0, HERE)
    import sharding_worker as sw
    flat = sw.table(n_rows)
    rows = sharding.table_rows(np.arange(n_total), n_rows, world, n_total)   # table reuse
    want = sw.fake_loglik({k: flat[k][rows] for k in ("ops", "brlen", "er", "pi", "alpha")})
    assert got["world"] == world
    np.testing.assert_array_equal(np.asarray(got["loglik"]), want)   # global sample order, nothing lost
    assert got["t_max"] == pytest.approx(0.010 * world)              # MAX over ranks
    assert got["n_rank0"] == sharding.shard_size(n_total, world, 0)


def test_failed_rank_fails_the_launch():
    status, out = sharding.spawn_ranks([sys.executable, WORKER, "8", "8", "1"], 2, timeout_s=120)
    assert status != 0

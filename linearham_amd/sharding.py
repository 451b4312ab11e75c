"""Multi-GPU partitioning of the phylo-HMM evaluation path: tree samples over ranks.

The iterations of PhyloHMM::RunPipeline's loop (src/PhyloHMM.cpp:414-442) are independent except for the
sampling RNG (src/HMM.cpp:56), so the evaluations shard with no data-path collective: sample i -> rank
i mod world, family constants replicated per rank by lh_family_create, and ONE gather
of the per-sample log-likelihoods to rank 0 per step.  This module is the only place that knows the
layout: bench.py, the host wrappers and tests/test_sharding_gloo.py all import it.

No numerics here; torch.distributed is plumbing (backend "nccl" is RCCL on ROCm, "gloo" goes through host
memory and is what the CPU tests and the fewer-GPUs-than-ranks rehearsal use).
"""
import os
import socket
import subprocess
import sys

import numpy as np


def shard_ids(n_total, world, rank):
    """Global sample numbers evaluated by `rank`: rank, rank + world, rank + 2 world, ... (< n_total)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return np.arange(rank, n_total, world, dtype=np.int64)


def shard_size(n_total, world, rank):
    return (n_total - rank + world - 1) // world if n_total > rank else 0


ROTATE = 977   # rows by which consecutive ranks' passes through a reused table are offset


def table_rows(global_ids, n_rows, world=1, n_total=None):
    """Row of the RevBayes table a global sample number reads, for a run of `n_total` samples over `world` ranks.
    A run that asks for no more samples than the table holds reads row g for sample g.  A weak-scaling run
    (n_total > n_rows) reuses the table: rank r's j-th sample (g = r + world j) reads row (j + ROTATE r) mod
    n_rows, so that every rank still works through distinct rows (as many as the table holds) instead of
    world copies of a world-th of them."""
    g = np.asarray(global_ids, dtype=np.int64)
    if world <= 1 or n_total is None or n_total <= n_rows:
        return g % n_rows
    return (g // world + (g % world) * ROTATE) % n_rows


def take_shard(flat, n_total, world, rank, keys=("ops", "brlen", "er", "pi", "alpha")):
    """This rank's slice of flattened per-sample input arrays (first axis = table row).
    Returns (dict of contiguous arrays, global sample numbers)."""
    ids = shard_ids(n_total, world, rank)
    rows = table_rows(ids, flat["n_rows"], world, n_total)
    return {k: np.ascontiguousarray(flat[k][rows]) for k in keys}, ids


def unshard(parts, n_total, world):
    """Inverse of shard_ids for the gathered result: parts[r][j] is the value of global sample r + j * world.
    Every part may be padded to the largest shard; padding is dropped."""
    out = np.empty(n_total, dtype=np.asarray(parts[0]).dtype)
    for r in range(world):
        m = shard_size(n_total, world, r)
        out[r::world] = np.asarray(parts[r])[:m]
    return out


def gather_loglik(local, n_total, world, rank, backend, device=None, out=None):
    """The path's single collective: every rank's log-likelihoods to rank 0.

    local: torch tensor (float64) of this rank's shard, on `device` for "nccl", anywhere for "gloo".
    Shards differ by at most one sample; each is padded to the largest so that the gather is regular.
    Returns, on rank 0, a torch tensor [world, max_shard] (the raw gather: use unshard() for sample order);
    None elsewhere.  `out` may hold a preallocated [world, max_shard] tensor (rank 0) to avoid
    allocation in timed loops."""
    import torch
    import torch.distributed as dist
    m = shard_size(n_total, world, 0)
    if world == 1:
        return local.reshape(1, -1)
    if backend == "nccl":
        src = local
        if src.numel() < m:
            src = torch.cat([src, torch.zeros(m - src.numel(), dtype=src.dtype, device=src.device)])
        if rank == 0:
            if out is None:
                out = torch.empty((world, m), dtype=src.dtype, device=src.device)
            dist.gather(src, gather_list=list(out.unbind(0)), dst=0)
            return out
        dist.gather(src, gather_list=None, dst=0)
        return None
    host = local.detach().cpu()
    if host.numel() < m:
        host = torch.cat([host, torch.zeros(m - host.numel(), dtype=host.dtype)])
    if rank == 0:
        parts = [torch.empty(m, dtype=host.dtype) for _ in range(world)]
        dist.gather(host, gather_list=parts, dst=0)
        res = torch.stack(parts)
        if out is not None:
            out.copy_(res)
            return out
        return res
    dist.gather(host, gather_list=None, dst=0)
    return None


def max_over_ranks(seconds, world, backend, device=None):
    """Slowest rank's time (the bench contract: MAX over ranks)."""
    if world == 1:
        return float(seconds)
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env():
    """(rank, local_rank, world) from the launcher's environment, or None when not launched as a rank."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        return int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"])), int(os.environ["WORLD_SIZE"])
    return None


def spawn_ranks(argv, world, timeout_s=None):
    """Start `world` rank processes of `argv` (one per GPU) as FRESH children of a parent that has not
    touched the GPU, wait for them and relay rank 0's stdout.  Returns (exit status, rank 0 stdout).

    The caller must not have initialised HIP (no torch.cuda.is_available(), no lh_* call): the children
    get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT exactly as torch.distributed.run would
    set them.  If any child fails, the others are terminated (by their own PIDs) and the status is
    nonzero; nothing is retried."""
    import time
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(world))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None))
    t0 = time.time()
    status = 0
    out0 = b""
    pending = set(range(world))
    import selectors
    sel = selectors.DefaultSelector()
    sel.register(procs[0].stdout, selectors.EVENT_READ)
    stdout_open = True
    while pending or stdout_open:
        if stdout_open:
            for key, _ in sel.select(timeout=0.2):
                chunk = os.read(key.fileobj.fileno(), 65536)
                if chunk:
                    out0 += chunk
                else:
                    sel.unregister(key.fileobj)
                    stdout_open = False
        else:
            time.sleep(0.2)
        for r in list(pending):
            rc = procs[r].poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 1
                    print("[launcher] rank %d exited with status %d; stopping the other ranks" % (r, rc),
                          file=sys.stderr, flush=True)
                    for q in pending:
                        procs[q].terminate()
        if timeout_s is not None and time.time() - t0 > timeout_s and pending:
            print("[launcher] timeout after %.0f s; stopping the ranks" % timeout_s, file=sys.stderr, flush=True)
            status = status or 124
            for q in pending:
                procs[q].kill()
            timeout_s = None
    return status, out0.decode(errors="replace")

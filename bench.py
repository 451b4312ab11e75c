#!/usr/bin/env python3
"""bench.py -- phylo-HMM log-likelihood evaluations per second on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (gamma rates, GTR eigen + P-matrices, Felsenstein pruning,
emission assembly, scaled forward sweep, log-likelihood) over one batch of tree samples of the
BASELINE.json configs[2] family (synthetic 100 leaves x 400 sites, 200 V / 30 D / 12 J alleles), with
the flattened inputs already resident in HBM.  The product path is: C++ host (linearham's PhyloHMM
surface, liblinearham_host.so) -> C ABI -> HIP kernels (liblinearham_hip.so).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Multi-GPU: tree samples shard across ranks (weak scaling: fixed batch per GPU), no data-path
collective inside an evaluation, one RCCL gather of the per-sample log-likelihoods per step.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 peak (MI355X_MICROARCH.md)
DEFAULT_BATCH = 24576          # tree samples per GPU per step (= one launch group of the C ABI)
WORKLOADS = {
    "config2": "BASELINE.json configs[2]: synthetic 100-leaf random tree, 400-site MSA, full V/D/J germline set "
               "(200 V / 30 D / 12 J alleles), R=4 rate categories",
    "config4": "BASELINE.json configs[4] shape on one GPU: synthetic 500-leaf random tree, 600-site MSA, full V/D/J "
               "germline set, R=4 rate categories (not the headline workload)",
    "small": "small synthetic family (development only, not the headline workload)",
}
PMC_PROFILE = "r01_v13_bench_pmc_per_launch.json"   # committed PMC passes of the default command


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(fam_dir, n_eval, budget_s):
    """Reference-algorithm CPU evaluation (oracle, dense) on the host cores of this box: a bounded
    sample of the same tree table.  Returns dict for the JSON line."""
    import numpy as np
    from oracle import linearham_oracle as orc
    from oracle import oracle_c
    from tests import desc_builder as db
    from tools import synth_family as sf
    oracle_c.build()
    t0 = time.time()
    h = orc.PhyloHMM(os.path.join(fam_dir, "cluster.yaml"), 0, os.path.join(fam_dir, "hmm_params"), 0)
    fam = oracle_c.COracleFamily(h, 4)
    rows = sf.read_trees_tsv(os.path.join(fam_dir, "trees.tsv"))
    log("[cpu_baseline] oracle family set-up %.1fs" % (time.time() - t0))
    cores = min(len(os.sched_getaffinity(0)), 16)   # the CPU share of a one-GPU box
    trees = [db.tree_arrays(orc.parse_newick(r["tree"]), h.xmsa_labels) for r in rows]

    def run(idx, threads):
        t = time.time()
        ll = fam.eval([trees[i] for i in idx], [rows[i]["er"] for i in idx], [rows[i]["pi"] for i in idx],
                      [rows[i]["alpha"] for i in idx], n_threads=threads)
        return time.time() - t, ll
    # single thread (the reference is single-threaded): 2 evaluations
    t1, ll1 = run([0, 1], 1)
    per_eval = t1 / 2
    n_all = int(max(cores, min(n_eval, budget_s * cores / max(per_eval, 1e-6))))
    n_all = max(cores, (n_all // cores) * cores)
    idx = [i % len(rows) for i in range(n_all)]
    tN, llN = run(idx, cores)
    return {"value": n_all / tN, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": "%d evaluations of the same tree table on %d threads (dense reference algorithm restated in C, "
                      "-O3 AVX2); single thread: %.3f evals/s" % (n_all, cores, 1.0 / per_eval),
            "single_thread_evals_per_s": 1.0 / per_eval}, {i: float(llN[k]) for k, i in enumerate(idx[:len(rows)])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH, help="tree samples per GPU per step")
    ap.add_argument("--preset", default="config2", choices=["config2", "config4", "small"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcie", action="store_true",
                    help="also time the host-pointer entry point (PCIe-inclusive rate; its launch groups are "
                         "sub-batches of 6144, so leave it off when profiling per-launch kernel durations)")
    ap.add_argument("--no-check", action="store_true", help="kernel timing experiments only")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (through host copies) only to rehearse the multi-rank path on "
                         "a box with fewer GPUs than ranks -- ranks then share devices (LOCAL_RANK modulo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    import linearham_amd
    from linearham_amd import host
    from tools import synth_family as sf

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")

    # ---- synthetic family in the reference's file formats (deterministic; same on every rank) -----
    spec = {"config2": sf.Spec(n_samples=256), "small": sf.Spec.small(n_samples=16),
            "config4": sf.Spec(n_leaves=500, n_sites=600, n_samples=64)}[args.preset]
    fam_dir = os.path.join(tempfile.gettempdir(), "lh_bench_%s_r%d" % (args.preset, rank))
    if not os.path.exists(os.path.join(fam_dir, "meta.json")):
        sf.generate(spec, fam_dir)

    # ---- product path: C++ host builds the family + flattens the tree table ---------------------------
    t0 = time.time()
    hmm = host.PhyloHMM(os.path.join(fam_dir, "cluster.yaml"), 0, os.path.join(fam_dir, "hmm_params"), 0)
    sizes = hmm.sizes()
    n = args.batch
    flat = hmm.flatten_tsv(os.path.join(fam_dir, "trees.tsv"), n)
    # different ranks evaluate different samples of the table (rotate by rank)
    roll = (rank * 37) % max(flat["n_rows"], 1)
    T, depth, R = flat["n_tips"], flat["max_depth"], 4
    d = {k: torch.from_numpy(np.roll(flat[k], roll, axis=0)).to(dev) for k in ("ops", "brlen", "er", "pi", "alpha")}
    loglik = torch.zeros(n, dtype=torch.float64, device=dev)
    gathered = torch.zeros(world * n, dtype=torch.float64, device=dev) if (world > 1 and rank == 0) else None
    lib = linearham_amd.load_library()
    fam_handle = flat["family"]
    import ctypes as C
    stream = torch.cuda.current_stream().cuda_stream
    log("[rank %d] host set-up %.2fs; %s; batch %d, max stack depth %d" % (rank, time.time() - t0, sizes, n, depth))

    def step():
        lib.check(lib.lib.lh_eval_batch_device(C.c_void_p(fam_handle), n, T, depth, d["ops"].data_ptr(),
                                               d["brlen"].data_ptr(), d["er"].data_ptr(), d["pi"].data_ptr(),
                                               d["alpha"].data_ptr(), R, loglik.data_ptr(), None,
                                               C.c_void_p(stream)))
        if world > 1:   # the single collective of the path: gather log-likelihoods on rank 0
            if args.backend == "nccl":
                dist.gather(loglik, gather_list=list(gathered.chunk(world)) if rank == 0 else None, dst=0)
            else:       # rehearsal: the same gather through host memory
                host_ll = loglik.cpu()
                parts = [torch.empty_like(host_ll) for _ in range(world)] if rank == 0 else None
                dist.gather(host_ll, gather_list=parts, dst=0)
                if rank == 0:
                    gathered.copy_(torch.cat(parts))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    lib.check(lib.lib.lh_profile_enable(C.c_void_p(fam_handle), 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms = [C.c_double() for _ in range(3)]
    groups = C.c_int64()
    lib.check(lib.lib.lh_profile_read(C.c_void_p(fam_handle), C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]),
                                      C.byref(groups)))
    lib.check(lib.lib.lh_profile_enable(C.c_void_p(fam_handle), 0))
    ll_host = loglik.cpu().numpy()
    n_bad = int(np.sum(~np.isfinite(ll_host)))
    if n_bad and not args.no_check:
        # configs[2] (the benchmark family) must be clean; the 500-leaf family has tree samples on which the
        # reference's own 2^(256*delta) equalisation overflows -- reported, not fatal
        if args.preset == "config2":
            raise SystemExit("non-finite log-likelihoods in the benchmark batch")
        log("[rank %d] %d of %d evaluations are non-finite (reference overflow behaviour)" % (rank, n_bad, n))

    if rank == 0:
        total_evals = world * n * args.steps
        value = total_evals / dt
        Cx, I = sizes["n_xmsa"], T - 2
        bytes_per_eval = Cx * (2 * I * R * 32 + T + 8)         # CLV-streaming model
        launches = max(groups.value, 1)
        prune_ms = ms[1].value / launches                        # average duration of one K1 launch
        per_launch = n * args.steps / launches                   # evaluations one launch group processes
        achieved = bytes_per_eval * per_launch / (prune_ms * 1e-3) / 1e9
        # HBM traffic of one K1 launch from the committed PMC passes of this same command (counters need
        # their own rocprofv3 runs; gfx950 correction: FETCH_SIZE counts wide coalesced reads at half).
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", PMC_PROFILE)
        if args.preset == "config2" and per_launch == DEFAULT_BATCH and os.path.exists(pmc_file):
            with open(pmc_file) as f:
                pmc = json.load(f)
            k1 = next((v for k, v in pmc.items() if "prune_kernel" in k), {})
            if "FETCH_SIZE" in k1 and "WRITE_SIZE" in k1:
                traffic = (2.0 * k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024.0
        # FP64 operations K1 has to do for this batch (the bound that actually applies): per site and rate
        # a cherry costs 4 multiplies, a tip-into-accumulator op 16 FMA + 4 mul, a pop-and-merge op
        # 32 FMA + 4 mul, the five-state close at the root 5 * (4 mul + 4 FMA).
        kinds = np.bincount((flat["ops"].reshape(-1, 4)[:, 0] & 15).astype(np.int64), minlength=3)[:3]
        flop_per_site_rate = (4 * kinds[0] + 36 * kinds[1] + 68 * kinds[2]) / float(flat["ops"].shape[0]) + 60
        n_pat, n_ucol = C.c_int32(), C.c_int32()
        lib.check(lib.lib.lh_family_info(C.c_void_p(fam_handle), C.byref(n_pat), C.byref(n_ucol)))
        k1_flops = flop_per_site_rate * n_pat.value * R * per_launch   # executed: identical columns are pruned once
        k1_tflops = k1_flops / (prune_ms * 1e-3) / 1e12
        out = {
            "metric": "phylo-HMM log-likelihood evals/sec (100-leaf x 400-site family)",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.preset],
                       "preset": args.preset, "tree_samples_per_gpu_per_step": n, "n_tips": T,
                       "n_sites": sizes["n_sites"], "xmsa_columns": Cx, "site_patterns": n_pat.value,
                       "distinct_xmsa_columns": n_ucol.value, "S_vd": sizes["s_vd"],
                       "S_dj": sizes["s_dj"], "W_vd": sizes["w_vd"], "W_dj": sizes["w_dj"],
                       "G": sizes["g_total"], "sharding": "tree samples over ranks; one RCCL gather of log-likelihoods"},
            "roofline": {"bound": "hbm", "kernel": "prune_kernel (K1, Felsenstein pruning)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "algorithmic_bytes_per_eval": bytes_per_eval, "evals_per_launch": per_launch,
                         "avg_launch_ms": prune_ms,
                         "fp64_valu": {"achieved": k1_tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": k1_tflops / FP64_VALU_PEAK_TFLOPS},
                         "note": "achieved = CLV-streaming model bytes (SURVEY 8(d)) / measured K1 time; the kernel "
                                 "keeps CLVs in registers and shares the tree across the naive states of a site, so it "
                                 "moves far fewer HBM bytes than the model and is FP64-VALU bound, see DESIGN.md"},
            "kernel_ms_per_step": {"model_K0": ms[0].value / args.steps, "prune_K1": ms[1].value / args.steps,
                                   "forward_K2": ms[2].value / args.steps,
                                   "launch_groups_per_step": launches / args.steps},
        }
        if world == 1 and args.pcie:
            # PCIe-inclusive rate through the host-pointer entry point (never `value`): H2D of the
            # flattened inputs, the same kernels, D2H of the log-likelihoods, synchronous per call.
            ll_pcie = np.zeros(n)
            p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
            h_ops, h_brl = np.ascontiguousarray(flat["ops"]), np.ascontiguousarray(flat["brlen"])
            reps = 3
            for rep in range(reps + 1):   # the first call sizes the pinned staging slots and is not timed
                if rep == 1:
                    t1 = time.perf_counter()
                lib.check(lib.lib.lh_eval_batch(C.c_void_p(fam_handle), n, T, depth, p(h_ops, C.c_int32),
                                                p(h_brl, C.c_double), p(flat["er"], C.c_double),
                                                p(flat["pi"], C.c_double), p(flat["alpha"], C.c_double), R,
                                                p(ll_pcie, C.c_double), None))
            out["pcie_inclusive_evals_per_s"] = n * reps / (time.perf_counter() - t1)
            if not args.no_check and not np.array_equal(ll_pcie, ll_host, equal_nan=True):
                raise SystemExit("host-pointer and device-pointer entry points disagree")
        if world == 1 and not args.no_cpu_baseline:
            base, ref_ll = cpu_baseline(fam_dir, 192, args.cpu_budget_s)
            out["cpu_baseline"] = {k: base[k] for k in ("value", "unit", "cores", "kind", "sample")}
            both = [(ll_host[i], v) for i, v in ref_ll.items() if i < n]
            if any(np.isfinite(g) != np.isfinite(v) for g, v in both):
                raise SystemExit("parity failure: GPU and CPU oracle disagree on which evaluations are finite")
            rel = max([abs(g - v) / abs(v) for g, v in both if np.isfinite(v)] or [0.0])
            out["delta_logl_vs_cpu_max_rel"] = rel
            out["speedup_vs_cpu_all_cores"] = value / base["value"]
            if rel > 1e-6:
                raise SystemExit("parity failure against the CPU oracle: max rel %.3e" % rel)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

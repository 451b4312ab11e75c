#!/usr/bin/env python3
"""bench.py -- phylo-HMM log-likelihood evaluations per second on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (gamma rates, GTR eigen + P-matrices, Felsenstein pruning,
emission assembly, scaled forward sweep, log-likelihood) over one batch of tree samples of the
BASELINE.json configs[2] family (synthetic 100 leaves x 400 sites, 200 V / 30 D / 12 J alleles), with
the flattened inputs already resident in HBM.  The product path is: C++ host (linearham's PhyloHMM
surface, liblinearham_host.so) -> C ABI -> HIP kernels (liblinearham_hip.so).

  python bench.py --gpus N --steps K --warmup W

Multi-GPU: `--gpus N` with no launcher environment starts its own N rank processes (one per GPU) as fresh
children before this process has touched the GPU, and relays rank 0's JSON line; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` each process is one rank.  Tree
samples shard over ranks (sample i -> rank i mod N, linearham_amd/sharding.py), no data-path collective inside
an evaluation, one RCCL gather of the per-sample log-likelihoods per step.  Prints ONE JSON line on rank 0.
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

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
# Vector FP64 peak: MI355X_MICROARCH.md lists the FP32 vector peak only (157.3 TFLOP/s = 256 CUs x 4 SIMDs x
# 32 lanes x 2 flop x 2.4 GHz); FP64 FMAs issue at half that rate (measured: tools/microbench/mfma_f64.hip
# sustains 63-72 TFLOP/s at the clock the chip holds under load), so the peak used here is 157.3 / 2.
FP64_VALU_PEAK_TFLOPS = 78.6
DEFAULT_BATCH = 49152          # tree samples per GPU per step (= one launch group of the C ABI)
WORKLOADS = {
    "config2": "BASELINE.json configs[2]: synthetic 100-leaf random tree, 400-site MSA, full V/D/J germline set "
               "(200 V / 30 D / 12 J alleles), R=4 rate categories",
    "config3": "BASELINE.json configs[3]: the configs[2] family, 10000 distinct tree samples (truth tree + 4 NNI moves, "
               "branch lengths x LogNormal(0,0.3)) sharded over the ranks (strong scaling: 10000 / N per GPU)",
    "config4": "BASELINE.json configs[4] shape: synthetic 500-leaf random tree, 600-site MSA, full V/D/J "
               "germline set, R=4 rate categories (not the headline workload)",
    "small": "small synthetic family (development only, not the headline workload)",
    "config2_ragged": "the configs[2] family with ragged reads: every sequence N-padded by 0-30 sites at either end, i.e. N "
                      "inside alignment columns (not the headline workload; its rate is the extra key mixed_n_evals_per_s)",
}
PMC_PROFILE = {"config2": "r04b_bench_pmc_per_launch.json", "config4": "r03_config4_pmc_per_launch.json"}
GEN_VERSION = 2                # bump when tools/synth_family.py changes what it writes


def log(*a):
    print(*a, file=sys.stderr, flush=True)


BRLEN_MEAN = None              # --brlen-mean (development runs on families with fewer site patterns)
DEV_LEAVES = None              # --leaves (development runs on smaller trees; only together with --brlen-mean)
LIVE_TRAFFIC = (None, None)    # (bytes per K1 launch, note) from live_k1_traffic(), collected before the GPU is touched


def preset_spec(preset, batch):
    from tools import synth_family as sf
    if preset == "config2" and BRLEN_MEAN is not None:
        return sf.Spec(n_samples=max(batch, 256), brlen_mean=BRLEN_MEAN, **({} if DEV_LEAVES is None else {"n_leaves": DEV_LEAVES}))
    if preset == "config2":     # the batch is drawn from >= batch distinct tree samples
        return sf.Spec(n_samples=max(batch, 256))
    if preset == "config2_ragged":   # (2048 distinct tree samples, cycled through the batch)
        return sf.Spec(n_samples=2048, ragged=30, **({} if BRLEN_MEAN is None else {"brlen_mean": BRLEN_MEAN}))
    if preset == "config3":
        return sf.Spec(n_samples=10000)
    if preset == "config4":
        return sf.Spec(n_leaves=500, n_sites=600, n_samples=max(batch, 64))
    return sf.Spec.small(n_samples=max(min(batch, 64), 16))


def family_dir(preset, spec):
    tag = "" if spec.brlen_mean == 0.01 else "_bl%g" % spec.brlen_mean
    if DEV_LEAVES is not None:
        tag += "_l%d" % DEV_LEAVES
    return os.path.join(tempfile.gettempdir(), "lh_bench_%s_n%d_v%d%s" % (preset, spec.n_samples, GEN_VERSION, tag))


def prepare_family(preset, batch, may_generate, wait_s=600):
    """The synthetic family in the reference's file formats (deterministic).  One process generates it
    (into a scratch directory, renamed into place when complete); the others wait for it."""
    from tools import synth_family as sf
    spec = preset_spec(preset, batch)
    d = family_dir(preset, spec)
    done = os.path.join(d, "meta.json")
    if os.path.exists(done):
        return d, spec
    if may_generate:
        tmp = "%s.tmp%d" % (d, os.getpid())
        t0 = time.time()
        sf.generate(spec, tmp)
        try:
            os.rename(tmp, d)
        except OSError:            # somebody else finished first
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
        log("[bench] generated %s (%d tree samples) in %.1fs" % (d, spec.n_samples, time.time() - t0))
        return d, spec
    t0 = time.time()
    while not os.path.exists(done):
        if time.time() - t0 > wait_s:
            raise SystemExit("timed out waiting for %s" % d)
        time.sleep(0.5)
    return d, spec


def usable_cores():
    """Cores this process may really use: its affinity mask, capped by the cgroup CPU quota (a one-GPU box
    shows every core of the host in the mask but grants a share of them)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = int(f.read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_oracle(fam_dir, row_ids, budget_s=None, n_timed=0):
    """Reference-algorithm CPU evaluation (oracle/, dense, plain C) of table rows `row_ids`; with
    n_timed > 0 also times a bounded sample on all host cores of this process's affinity mask.
    Returns ({row: loglik}, baseline dict or None)."""
    from oracle import linearham_oracle as orc
    from oracle import oracle_c
    from tests import desc_builder as db
    from tools import synth_family as sf
    oracle_c.build()
    t0 = time.time()
    h = orc.PhyloHMM(os.path.join(fam_dir, "cluster.yaml"), 0, os.path.join(fam_dir, "hmm_params"), 0)
    fam = oracle_c.COracleFamily(h, 4)
    rows = sf.read_trees_tsv(os.path.join(fam_dir, "trees.tsv"), max_rows=max(max(row_ids) + 1, n_timed, 2))
    log("[cpu oracle] family set-up %.1fs" % (time.time() - t0))
    cores = usable_cores()
    cache = {}

    def tree(i):
        if i not in cache:
            cache[i] = db.tree_arrays(orc.parse_newick(rows[i]["tree"]), h.xmsa_labels)
        return cache[i]

    def run(idx, threads):
        t = time.time()
        ll = fam.eval([tree(i) for i in idx], [rows[i]["er"] for i in idx], [rows[i]["pi"] for i in idx],
                      [rows[i]["alpha"] for i in idx], n_threads=threads)
        return time.time() - t, ll
    base = None
    ref = {}
    if n_timed > 0:
        t1, ll1 = run([0, 1], 1)     # single thread (the reference program is single-threaded)
        per_eval = t1 / 2
        n_all = int(max(cores, min(n_timed, budget_s * cores / max(per_eval, 1e-6))))
        n_all = max(cores, (n_all // cores) * cores)
        idx = [i % len(rows) for i in range(n_all)]
        for i in set(idx):
            tree(i)
        tN, llN = run(idx, cores)
        ref.update({i: float(llN[k]) for k, i in enumerate(idx)})
        base = {"value": n_all / tN, "unit": "evals/s", "cores": cores, "kind": "port",
                "sample": "%d evaluations (the first rows of the same tree table) on %d threads = every core this "
                          "process may use (affinity mask capped by the cgroup CPU quota); dense reference algorithm restated in C (oracle/oracle_kernels.c, "
                          "-O3 AVX2; upstream builds without -O); single thread: %.3f evals/s; CPU: %s"
                          % (n_all, cores, 1.0 / per_eval, cpu_model()),
                "single_thread_evals_per_s": 1.0 / per_eval, "cpu_model": cpu_model()}
    todo = [i for i in row_ids if i not in ref]
    if todo:
        _, ll = run(todo, cores)
        ref.update({i: float(v) for i, v in zip(todo, ll)})
    return ref, base


def extra_rates(args, lib, fam_handle, hmm, d, T, depth, R, n, dev, stream, fam_dir, sizes, shard):
    """Two more rates of the same family, measured OUTSIDE the `value` region (never part of it), inputs resident:
    pipeline_rows_per_s -- what RunPipeline's device side does per row: K0-K2 with the forward arrays written, then the
    naive-sequence state draws on them (K4, lh_eval_sample_batch_device; src/HMM.cpp:358-431);
    asr_tree_samples_per_s -- the ancestral-sequence step (K3, lh_asr_batch_device; scripts/run_bootstrap_asr_ess.R:48-104),
    batch 4096, its draws checked against oracle/asr_oracle.py on two samples (that oracle's parity is UNPINNED: the
    reference holds no fixture for the step)."""
    import ctypes as C
    import numpy as np
    import torch
    fam = C.c_void_p(fam_handle)
    res = {}
    # ---- pipeline rows: evaluation + device sampler ----
    n_words = lib.lib.lh_sample_words(fam)
    n_states = lib.lib.lh_sample_states(fam)
    if n_words > 0:
        g = torch.Generator(device="cpu").manual_seed(7)
        words = torch.randint(0, 2 ** 31 - 1, (n, n_words), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
        states = torch.zeros((n, n_states), dtype=torch.int32, device=dev)
        ll2 = torch.zeros(n, dtype=torch.float64, device=dev)

        def row_step():
            lib.check(lib.lib.lh_eval_sample_batch_device(fam, n, T, depth, d["ops"].data_ptr(), d["brlen"].data_ptr(),
                                                          d["er"].data_ptr(), d["pi"].data_ptr(), d["alpha"].data_ptr(), R,
                                                          words.data_ptr(), ll2.data_ptr(), None, states.data_ptr(),
                                                          C.c_void_p(stream)))
        row_step()
        torch.cuda.synchronize()
        k = max(1, min(args.steps, 5))
        t = time.perf_counter()
        for _ in range(k):
            row_step()
        torch.cuda.synchronize()
        res["pipeline_rows_per_s"] = n * k / (time.perf_counter() - t)
        res["pipeline_rows_note"] = ("K0-K2 with forward arrays + K4 (device sampler) on resident inputs, %d rows per step; "
                                     "parsing, formatting and file I/O of RunPipeline are host work outside this figure" % n)
        if not args.no_check and int(states.min().item()) < 0:
            raise SystemExit("device sampler returned a negative state index")
    # ---- ancestral-sequence step ----
    try:
        m = min(n, 4096)
        L = sizes["n_sites"]
        rates = torch.zeros((m, R), dtype=torch.float64, device=dev)
        from linearham_amd.capi import _EvalOutputs
        outs = _EvalOutputs()
        outs.rates = C.cast(rates.data_ptr(), C.POINTER(C.c_double))
        ll3 = torch.zeros(m, dtype=torch.float64, device=dev)
        lib.check(lib.lib.lh_eval_batch_device(fam, m, T, depth, d["ops"].data_ptr(), d["brlen"].data_ptr(),
                                               d["er"].data_ptr(), d["pi"].data_ptr(), d["alpha"].data_ptr(), R,
                                               ll3.data_ptr(), C.byref(outs), C.c_void_p(stream)))
        rng = np.random.default_rng(1)
        naive = rng.integers(0, 4, size=(m, L)).astype(np.uint8)
        d_naive = torch.from_numpy(naive).to(dev)
        anc = torch.zeros((m, T - 2, L), dtype=torch.uint8, device=dev)

        def asr_step(seed):
            lib.check(lib.lib.lh_asr_batch_device(fam, m, T, depth, d["ops"].data_ptr(), d["brlen"].data_ptr(),
                                                  d["er"].data_ptr(), d["pi"].data_ptr(), rates.data_ptr(), R,
                                                  d_naive.data_ptr(), seed, 0, anc.data_ptr(), None, C.c_void_p(stream)))
        asr_step(1)
        torch.cuda.synchronize()
        k = max(1, min(args.steps, 5))
        t = time.perf_counter()
        for s_ in range(k):
            asr_step(100 + s_)
        torch.cuda.synchronize()
        res["asr_tree_samples_per_s"] = m * k / (time.perf_counter() - t)
        res["asr_note"] = ("ancestral-sequence sampling (K3) at batch %d, unmixed K1 planes + K3a/K3s/K3b per step; "
                           "checked against oracle/asr_oracle.py, whose parity is UNPINNED (no reference fixture)" % m)
        if not args.no_check:
            from oracle import asr_oracle as ao
            from oracle import linearham_oracle as orc
            from linearham_amd import host as _host
            from tools import synth_family as sf
            o = orc.PhyloHMM(os.path.join(fam_dir, "cluster.yaml"), 0, os.path.join(fam_dir, "hmm_params"), 0)
            labels = list(o.xmsa_labels)
            rows = sf.read_trees_tsv(os.path.join(fam_dir, "trees.tsv"), max_rows=2)
            a = anc[:2].cpu().numpy()
            r_host = rates[:2].cpu().numpy()
            for i in range(2):
                children, root, brlen = _host.newick_arrays(rows[i]["tree"], labels)
                _, anc_ref, _ = ao.asr_sample(children, root, brlen, T, o.msa, naive[i], rows[i]["er"], np.asarray(rows[i]["pi"]),
                                              r_host[i], 100 + k - 1, i)
                if int((anc_ref != a[i]).sum()):
                    raise SystemExit("parity failure: ancestral-sequence draws differ from oracle/asr_oracle.py")
            res["asr_samples_checked"] = 2
    except RuntimeError as e:
        # a tree too large for K3's LDS tables: the step does not exist for this shape; anything else is a failure of
        # the product path and must not vanish from the line
        if "too large for the sampling kernel" not in str(e) and "too many tips" not in str(e):
            raise
        res["asr_note"] = "not run: %s" % e
    # ---- the same workload with N inside alignment columns (ragged reads) ----
    if args.preset == "config2" and not args.no_mixed_n:
        res.update(mixed_n_rate(args, lib, T, R, n, dev, stream))
    return res


def mixed_n_rate(args, lib, T, R, n, dev, stream):
    """mixed_n_evals_per_s: K0-K2 on the configs[2] family with RAGGED READS (every sequence N-padded by 0-30 sites at
    either end: N inside alignment columns, libpll's 1111 tips -- src/HMM.cpp:69-83, src/PhyloHMM.cpp:368-370), inputs
    resident, measured OUTSIDE the `value` region.  Such alignments run K1's N-aware instantiation (row sums of a tip's
    matrix formed on the spot; no assembly walk); two rows are checked against the dense C oracle."""
    import ctypes as C
    import numpy as np
    import torch
    from linearham_amd import host
    fam_dir, spec = prepare_family("config2_ragged", n, may_generate=True)
    hmm = host.PhyloHMM(os.path.join(fam_dir, "cluster.yaml"), 0, os.path.join(fam_dir, "hmm_params"), 0)
    flat = hmm.flatten_tsv(os.path.join(fam_dir, "trees.tsv"), n)
    assert flat["n_tips"] == T
    d = {k: torch.from_numpy(np.ascontiguousarray(flat[k][:n])).to(dev) for k in ("ops", "brlen", "er", "pi", "alpha")}
    ll = torch.zeros(n, dtype=torch.float64, device=dev)
    fam = C.c_void_p(flat["family"])

    def step():
        lib.check(lib.lib.lh_eval_batch_device(fam, n, T, flat["max_depth"], d["ops"].data_ptr(), d["brlen"].data_ptr(),
                                               d["er"].data_ptr(), d["pi"].data_ptr(), d["alpha"].data_ptr(), R,
                                               ll.data_ptr(), None, C.c_void_p(stream)))
    step()
    torch.cuda.synchronize()
    lib.check(lib.lib.lh_profile_enable(fam, 1))
    k = max(1, min(args.steps, 5))
    t = time.perf_counter()
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    ms = [C.c_double() for _ in range(3)]
    groups = C.c_int64()
    lib.check(lib.lib.lh_profile_read(fam, C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]), C.byref(groups)))
    lib.check(lib.lib.lh_profile_enable(fam, 0))
    n_pat = C.c_int32()
    lib.check(lib.lib.lh_family_info(fam, C.byref(n_pat), None))
    res = {"mixed_n_evals_per_s": n * k / dt,
           "mixed_n_note": "configs[2] with ragged reads (N inside alignment columns; tools/synth_family.py ragged=30), %d tree "
                           "samples per step (%d distinct), %d site patterns; K1 form %s, K1 %.3f ms per step; outside `value`"
                           % (n, flat["n_rows"], n_pat.value, lib.lib.lh_family_prune_form(fam).decode(), ms[1].value / k)}
    if not args.no_check:
        got = ll[:2].cpu().numpy()
        ref, _ = cpu_oracle(fam_dir, [0, 1])
        rel = max(abs(got[i] - ref[i]) / abs(ref[i]) for i in (0, 1))
        res["mixed_n_delta_logl_vs_cpu_max_rel"] = rel
        if not (rel <= 1e-6):
            raise SystemExit("parity failure on the ragged-read family against the CPU oracle: max rel %.3e" % rel)
    return res


def live_k1_traffic(args):
    """HBM-side bytes of one K1 launch, measured NOW on this box: FETCH_SIZE and WRITE_SIZE of the pruning kernel from two short
    child runs of this script under `rocprofv3 --pmc` (counters cannot be collected inside the measuring process, and the two
    do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).  Runs BEFORE this process touches the GPU.  Returns
    (bytes per launch with the gfx950 factor 2 x FETCH + WRITE, note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    per_launch = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="lh_pmc_")
        cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "run", "--", sys.executable, os.path.abspath(__file__),
               "--preset", args.preset, "--batch", str(args.batch), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
               "--no-check", "--no-extras", "--no-forward-rate", "--no-live-pmc"]
        if BRLEN_MEAN is not None:
            cmd += ["--brlen-mean", repr(BRLEN_MEAN)]
        try:
            r = subprocess.run(cmd, cwd=tempfile.gettempdir(), env=dict(os.environ, TMPDIR=tempfile.gettempdir()),
                               capture_output=True, text=True, timeout=300)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (ctr, r.returncode)
            vals = []
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if "prune_kernel" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, "no prune_kernel dispatch in the %s pass" % ctr
            per_launch[ctr] = sum(vals) / len(vals)
        except (OSError, subprocess.SubprocessError, ValueError, KeyError) as e:
            return None, "rocprofv3 --pmc %s: %s" % (ctr, e)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    total = (2.0 * per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"]) * 1024.0
    return total, ("measured in this run: two child passes of `rocprofv3 --pmc` over 3 launches of the same workload "
                   "(FETCH_SIZE %.0f KiB, WRITE_SIZE %.0f KiB per launch as reported; bytes = 2 x FETCH + WRITE, the gfx950 "
                   "correction of MI355X_MICROARCH.md)" % (per_launch["FETCH_SIZE"], per_launch["WRITE_SIZE"]))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="tree samples per GPU per step (weak-scaling presets)")
    ap.add_argument("--preset", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--brlen-mean", type=float, default=None,
                    help="development only (not the headline workload): mean branch length of the synthetic truth tree "
                         "(default 0.01: 253 site patterns for config2; 0.002 gives a family with ~100 patterns)")
    ap.add_argument("--leaves", type=int, default=None,
                    help="development only, with --brlen-mean: leaves of the synthetic tree (default 100)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcie", action="store_true",
                    help="also time the host-pointer entry point (PCIe-inclusive rate; its launch groups are "
                         "sub-batches of 6144, so leave it off when profiling per-launch kernel durations)")
    ap.add_argument("--no-check", action="store_true", help="kernel timing experiments only")
    ap.add_argument("--no-forward-rate", action="store_true", help="skip the second timed loop (forward arrays on)")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--cpu-samples", type=int, default=192,
                    help="evaluations of the CPU baseline sample (also the rows the parity check covers)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra keys measured outside the `value` region (pipeline_rows_per_s, asr_tree_samples_per_s)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not spawn the two short `rocprofv3 --pmc` child runs that measure K1's HBM traffic for roofline.traffic "
                         "(the figure then comes from the committed profile and says so)")
    ap.add_argument("--no-mixed-n", action="store_true",
                    help="skip the extra key mixed_n_evals_per_s (the configs[2] family with ragged reads, outside `value`)")
    ap.add_argument("--timeout-s", type=float, default=1500.0,
                    help="--gpus N without a launcher: stop the rank processes after this many seconds (exit status 124)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (through host copies) only to rehearse the multi-rank path on "
                         "a box with fewer GPUs than ranks -- ranks then share devices (LOCAL_RANK modulo)")
    args = ap.parse_args()
    global BRLEN_MEAN, DEV_LEAVES
    BRLEN_MEAN = args.brlen_mean
    DEV_LEAVES = args.leaves if args.brlen_mean is not None else None
    if args.batch is None:
        args.batch = {"config2": DEFAULT_BATCH, "config2_ragged": DEFAULT_BATCH, "config3": 0, "config4": 6144, "small": 64}[args.preset]
    return args


def launcher(args):
    """--gpus N without a launcher environment: this process stays off the GPU, prepares the input files,
    starts N fresh rank processes and relays rank 0's JSON line."""
    from linearham_amd import sharding
    prepare_family(args.preset, args.batch, may_generate=True)
    argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    status, out = sharding.spawn_ranks(argv, args.gpus, timeout_s=args.timeout_s)
    sys.stdout.write(out)
    sys.stdout.flush()
    if status == 0 and not any(l.startswith("{") for l in out.splitlines()):
        log("[launcher] rank 0 printed no JSON line")
        status = 1
    return status


def worker(args, rank, local_rank, world):
    import ctypes as C
    import numpy as np
    import torch
    import torch.distributed as dist
    import linearham_amd
    from linearham_amd import host, sharding
    from linearham_amd.capi import _EvalOutputs

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.backend == "gloo":
        local_rank %= n_dev
    elif local_rank >= n_dev:
        raise SystemExit("rank %d has no GPU (%d visible); use --backend gloo to rehearse on fewer GPUs" % (rank, n_dev))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")

    # ---- synthetic family in the reference's file formats (deterministic; one copy per box) -------
    fam_dir, spec = prepare_family(args.preset, args.batch, may_generate=(rank == 0))

    # ---- product path: C++ host builds the family + flattens the tree table -----------------------
    t0 = time.time()
    hmm = host.PhyloHMM(os.path.join(fam_dir, "cluster.yaml"), 0, os.path.join(fam_dir, "hmm_params"), 0)
    sizes = hmm.sizes()
    strong = args.preset == "config3"
    n_total = spec.n_samples if strong else world * args.batch
    # this rank's tree samples (global sample g -> rank g mod N) and the table rows they read: only those rows are
    # parsed and scheduled here
    ids = sharding.shard_ids(n_total, world, rank)
    my_rows = sharding.table_rows(ids, spec.n_samples, world, n_total)
    n = len(ids)
    flat = hmm.flatten_tsv(os.path.join(fam_dir, "trees.tsv"), n, rows=my_rows if n else [0])
    if flat["n_rows"] != spec.n_samples:
        raise SystemExit("the tree table has %d rows, expected %d" % (flat["n_rows"], spec.n_samples))
    shard = {k: np.ascontiguousarray(flat[k][:n]) for k in ("ops", "brlen", "er", "pi", "alpha")}
    T, depth, R = flat["n_tips"], flat["max_depth"], 4
    d = {k: torch.from_numpy(v).to(dev) for k, v in shard.items()}
    loglik = torch.zeros(max(n, 1), dtype=torch.float64, device=dev)[:n]
    m0 = sharding.shard_size(n_total, world, 0)
    gathered = (torch.zeros((world, m0), dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
                if (world > 1 and rank == 0) else None)
    lib = linearham_amd.load_library()
    fam_handle = flat["family"]
    stream = torch.cuda.current_stream().cuda_stream
    log("[rank %d of %d] host set-up %.2fs; %s; %d of %d samples (%d distinct table rows parsed here, table of %d), "
        "max stack depth %d" % (rank, world, time.time() - t0, sizes, n, n_total, len(set(my_rows.tolist())), flat["n_rows"], depth))

    def evaluate(outs=None):
        lib.check(lib.lib.lh_eval_batch_device(C.c_void_p(fam_handle), n, T, depth, d["ops"].data_ptr(),
                                               d["brlen"].data_ptr(), d["er"].data_ptr(), d["pi"].data_ptr(),
                                               d["alpha"].data_ptr(), R, loglik.data_ptr(), outs,
                                               C.c_void_p(stream)))

    def step(outs=None):
        evaluate(outs)
        # the single collective of the path: every rank's log-likelihoods to rank 0
        return sharding.gather_loglik(loglik, n_total, world, rank, args.backend, dev, out=gathered)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(k, outs=None):
        barrier()
        t = time.perf_counter()
        res = None
        for _ in range(k):
            res = step(outs)
        barrier()
        return sharding.max_over_ranks(time.perf_counter() - t, world, args.backend, dev), res

    for _ in range(args.warmup):
        step()
    barrier()
    lib.check(lib.lib.lh_profile_enable(C.c_void_p(fam_handle), 1))
    dt, res = timed(args.steps)
    ms = [C.c_double() for _ in range(3)]
    groups = C.c_int64()
    lib.check(lib.lib.lh_profile_read(C.c_void_p(fam_handle), C.byref(ms[0]), C.byref(ms[1]), C.byref(ms[2]),
                                      C.byref(groups)))
    lib.check(lib.lib.lh_profile_enable(C.c_void_p(fam_handle), 0))
    ll_host = loglik.cpu().numpy()
    ll_all = sharding.unshard(res.cpu().numpy(), n_total, world) if rank == 0 else None

    # the same step with the arrays SampleNaiveSequence consumes (forward rows + ScaleMatrix counts) written
    dt_fwd = None
    if not args.no_forward_rate and n > 0:
        fsize, ssize = lib.lib.lh_forward_size(C.c_void_p(fam_handle)), lib.lib.lh_scaler_size(C.c_void_p(fam_handle))
        fwd = torch.empty((n, fsize), dtype=torch.float64, device=dev)
        sco = torch.empty((n, ssize), dtype=torch.int32, device=dev)
        outs = _EvalOutputs()
        outs.forward = C.cast(fwd.data_ptr(), C.POINTER(C.c_double))
        outs.scaler_counts = C.cast(sco.data_ptr(), C.POINTER(C.c_int32))
        ll_before = loglik.clone()
        step(C.byref(outs))
        k_fwd = max(1, min(args.steps, 10))
        dt_fwd, _ = timed(k_fwd, C.byref(outs))
        dt_fwd /= k_fwd
        if not args.no_check and not torch.equal(torch.nan_to_num(ll_before), torch.nan_to_num(loglik)):
            raise SystemExit("log-likelihoods changed when the forward arrays were requested")
        del fwd, sco

    n_bad = int(np.sum(~np.isfinite(ll_host)))
    if n_bad and not args.no_check:
        # configs[2] (the benchmark family) must be clean; the 500-leaf family has tree samples on which the
        # reference's own 2^(256*delta) equalisation overflows -- reported, not fatal
        if args.preset in ("config2", "config3"):
            raise SystemExit("non-finite log-likelihoods in the benchmark batch")
        log("[rank %d] %d of %d evaluations are non-finite (reference overflow behaviour)" % (rank, n_bad, n))

    if rank == 0:
        total_evals = n_total * args.steps
        value = total_evals / dt
        Cx = sizes["n_xmsa"]
        launches = max(groups.value, 1)
        prune_ms = ms[1].value / launches                        # average duration of one K1 launch
        per_launch = n * args.steps / launches                   # evaluations one launch group processes
        n_pat, n_ucol = C.c_int32(), C.c_int32()
        lib.check(lib.lib.lh_family_info(C.c_void_p(fam_handle), C.byref(n_pat), C.byref(n_ucol)))
        # FP64 operations K1 executes for one (pattern, rate) of one evaluation, counted strictly (multiply or
        # add = 1, FMA = 2) from linearham_amd/csrc/lh_prune.hip: a 4x4 mat-vec = 4 mul + 12 FMA = 28; every op
        # ends in the element-wise product (4 mul): cherry 4, tip-into-accumulator 28 + 4 = 32, pop-and-merge
        # 2 * 28 + 4 = 60; the close at the root: 4 mul (pi * clv), five states x (1 mul + 3 FMA) = 35, and the
        # row sums the N state needs (12 adds) = 51.  Identical alignment columns are pruned once (n_pat).
        kinds = np.bincount((shard["ops"].reshape(-1, 4)[:, 0] & 15).astype(np.int64), minlength=3)[:3]
        flop_per_site_rate = (4 * kinds[0] + 32 * kinds[1] + 60 * kinds[2]) / float(max(n, 1)) + 51
        k1_flops = flop_per_site_rate * n_pat.value * R * per_launch
        k1_tflops = k1_flops / (prune_ms * 1e-3) / 1e12
        # HBM traffic of one K1 launch from the committed PMC passes of this same command (counters need
        # their own rocprofv3 runs; gfx950 correction: FETCH_SIZE counts wide coalesced reads at half).
        traffic, traffic_note = LIVE_TRAFFIC if world == 1 and per_launch == args.batch else (None, None)
        pmc_file = os.path.join(ROOT, "profiles", PMC_PROFILE.get(args.preset, "-"))
        if traffic is None and world == 1 and per_launch == args.batch and os.path.exists(pmc_file) and BRLEN_MEAN is None:
            with open(pmc_file) as f:
                pmc = json.load(f)
            if pmc.get("_evals_per_launch", per_launch) == per_launch:
                k1 = next((v for k, v in pmc.items() if "prune_kernel" in k), {})
                if "FETCH_SIZE" in k1 and "WRITE_SIZE" in k1:
                    traffic = (2.0 * k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024.0
                    traffic_note = ("NOT measured by this run (%s): profiles/%s, HBM-side bytes per K1 launch from separate "
                                    "rocprofv3 --pmc passes of this same command (2 x FETCH_SIZE + WRITE_SIZE)"
                                    % (traffic_note or "live counters off", PMC_PROFILE.get(args.preset)))
        model_bytes_per_eval = Cx * (2 * (T - 2) * R * 32 + T + 8)   # CLV-streaming model
        out = {
            "metric": "phylo-HMM log-likelihood evals/sec (100-leaf x 400-site family)",
            "value": value, "unit": "evals/s", "n_gpus": dist.get_world_size() if world > 1 else 1,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.preset],
                       "preset": args.preset, "tree_samples_per_step": n_total,
                       "tree_samples_per_gpu_per_step": n, "distinct_tree_samples": min(flat["n_rows"], n_total),
                       "distinct_tree_samples_per_gpu": int(len(set(sharding.table_rows(ids, flat["n_rows"], world, n_total).tolist()))),
                       "n_tips": T, "n_sites": sizes["n_sites"], "xmsa_columns": Cx, "site_patterns": n_pat.value,
                       "k1_form": lib.lib.lh_family_prune_form(C.c_void_p(fam_handle)).decode(),
                       "distinct_xmsa_columns": n_ucol.value, "S_vd": sizes["s_vd"],
                       "S_dj": sizes["s_dj"], "W_vd": sizes["w_vd"], "W_dj": sizes["w_dj"],
                       "G": sizes["g_total"], "backend": dist.get_backend() if world > 1 else None,
                       "collective_world_size": dist.get_world_size() if world > 1 else 1,
                       "sharding": "tree sample i -> rank i mod N; one RCCL gather of log-likelihoods per step"},
            "roofline": {"bound": "fp64_valu", "kernel": "prune_kernel (K1, Felsenstein pruning)",
                         "achieved": k1_tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": k1_tflops / FP64_VALU_PEAK_TFLOPS,
                         "traffic": traffic,
                         "traffic_source": traffic_note if traffic else None,
                         "hbm_gbs": (traffic / (prune_ms * 1e-3) / 1e9) if traffic else None,
                         "flop_per_launch": k1_flops, "evals_per_launch": per_launch, "avg_launch_ms": prune_ms,
                         "peak_source": "half the 157.3 TFLOP/s FP32 vector peak of MI355X_MICROARCH.md (FP64 FMAs "
                                        "issue at half rate; the guide lists no FP64 figure)",
                         "note": "strict flop count (mul/add 1, FMA 2) of the pruning arithmetic per (site pattern, rate) as the "
                                 "schedule states it (lh_prune.hip's register-stack walk executes exactly these; the cherry-table "
                                 "form that fused shapes run since round 4 folds a quarter of the ops into table look-ups and executes "
                                 "fewer: `achieved` prices the ALGORITHM's flops against the launch time); the kernel keeps CLVs in "
                                 "registers, so HBM is not its bound: "
                                 "SURVEY 8(d)'s CLV-streaming model (%d B per evaluation) would need %.0f GB/s at "
                                 "this launch time -- a statement about the model, not about the kernel"
                                 % (model_bytes_per_eval, model_bytes_per_eval * per_launch / (prune_ms * 1e-3) / 1e9)},
            "kernel_ms_per_step": {"model_K0": ms[0].value / args.steps, "prune_K1": ms[1].value / args.steps,
                                   "forward_K2": ms[2].value / args.steps,
                                   "launch_groups_per_step": launches / args.steps},
        }
        if dt_fwd is not None:
            out["evals_per_s_with_forward"] = n_total / dt_fwd
            out["with_forward_note"] = ("the same step with lh_eval_outputs.forward and .scaler_counts written (what "
                                        "SampleNaiveSequence consumes); `value` leaves them off, as SURVEY 8(d) excludes sampling")
        if world == 1 and not args.no_extras and n > 0:
            out.update(extra_rates(args, lib, fam_handle, hmm, d, T, depth, R, n, dev, stream, fam_dir, sizes, shard))
        if world == 1 and args.pcie:
            # PCIe-inclusive rate through the host-pointer entry point (never `value`): H2D of the
            # flattened inputs, the same kernels, D2H of the log-likelihoods, synchronous per call.
            ll_pcie = np.zeros(n)
            p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
            reps = 3
            for rep in range(reps + 1):   # the first call sizes the pinned staging slots and is not timed
                if rep == 1:
                    t1 = time.perf_counter()
                lib.check(lib.lib.lh_eval_batch(C.c_void_p(fam_handle), n, T, depth, p(shard["ops"], C.c_int32),
                                                p(shard["brlen"], C.c_double), p(shard["er"], C.c_double),
                                                p(shard["pi"], C.c_double), p(shard["alpha"], C.c_double), R,
                                                p(ll_pcie, C.c_double), None))
            out["pcie_inclusive_evals_per_s"] = n * reps / (time.perf_counter() - t1)
            if not args.no_check and not np.array_equal(ll_pcie, ll_host, equal_nan=True):
                raise SystemExit("host-pointer and device-pointer entry points disagree")
        if not args.no_check:
            # parity against the CPU oracle: at N = 1 on the timed baseline sample, at N > 1 on a few samples of
            # every rank's shard (global sample g was evaluated by rank g mod N)
            want_base = world == 1 and not args.no_cpu_baseline
            check_ids = list(range(min(n_total, 2 * world if world > 1 else 2)))
            rows = [int(r) for r in sharding.table_rows(check_ids, flat["n_rows"], world, n_total)]
            if want_base or world > 1 or args.preset in ("small", "config3"):
                ref_ll, base = cpu_oracle(fam_dir, rows, args.cpu_budget_s, args.cpu_samples if want_base else 0)
                if base:
                    out["cpu_baseline"] = {k: base[k] for k in ("value", "unit", "cores", "kind", "sample")}
                    out["speedup_vs_cpu_all_cores"] = value / base["value"]
                    check_ids = [g for g in range(min(n_total, flat["n_rows"])) if g in ref_ll]
                both = [(ll_all[g], ref_ll[int(sharding.table_rows([g], flat["n_rows"], world, n_total)[0])]) for g in check_ids]
                if any(np.isfinite(g) != np.isfinite(v) for g, v in both):
                    raise SystemExit("parity failure: GPU and CPU oracle disagree on which evaluations are finite")
                rel = max([abs(g - v) / abs(v) for g, v in both if np.isfinite(v)] or [0.0])
                out["delta_logl_vs_cpu_max_rel"] = rel
                out["delta_logl_samples_checked"] = len(both)
                if rel > 1e-6:
                    raise SystemExit("parity failure against the CPU oracle: max rel %.3e" % rel)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    from linearham_amd import sharding
    env = sharding.rank_env()
    if env is None:
        if args.gpus > 1:
            return launcher(args)
        env = (0, 0, 1)
    rank, local_rank, world = env
    global LIVE_TRAFFIC
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and not args.no_live_pmc and not under_profiler and args.preset in ("config2", "config4"):
        t0 = time.time()
        LIVE_TRAFFIC = live_k1_traffic(args)
        log("[bench] K1 HBM traffic from live rocprofv3 passes: %s (%.0f s)" % (LIVE_TRAFFIC, time.time() - t0))
    if world != args.gpus:
        log("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    return worker(args, rank, local_rank, world)


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Throughput of the ancestral-sequence sampling step (K3, lh_asr_batch_device) on the configs[2] family,
inputs resident in HBM.  Not the headline metric (bench.py is); prints one JSON line.

  python bench_asr.py [--batch 2048] [--steps 5] [--warmup 1] [--preset config2|config4]

Like bench.py, the only use of oracle/ here is the `cpu_baseline` leg (the checker timed beside the kernel); the
measured path and its inputs come from the product (C++ host + C ABI; the site rates are K0's).

K3 is the CLV-streaming kernel of SURVEY 8(d)'s byte model: per tree sample it writes and reads back
32 B per (inner node, site) plus one state byte per (inner node, site)."""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--preset", default="config2", choices=["config2", "config4", "small"])
    args = ap.parse_args()
    import numpy as np
    import torch
    import linearham_amd
    from linearham_amd import host
    from tools import synth_family as sf
    dev = torch.device("cuda", 0)
    spec = {"config2": sf.Spec(n_samples=256), "small": sf.Spec.small(n_samples=16),
            "config4": sf.Spec(n_leaves=500, n_sites=600, n_samples=64)}[args.preset]
    fam_dir = os.path.join(tempfile.gettempdir(), "lh_bench_%s_r0" % args.preset)
    if not os.path.exists(os.path.join(fam_dir, "meta.json")):
        sf.generate(spec, fam_dir)
    hmm = host.PhyloHMM(os.path.join(fam_dir, "cluster.yaml"), 0, os.path.join(fam_dir, "hmm_params"), 0)
    sizes = hmm.sizes()
    n = args.batch
    flat = hmm.flatten_tsv(os.path.join(fam_dir, "trees.tsv"), n)
    T, depth, R, L = flat["n_tips"], flat["max_depth"], 4, sizes["n_sites"]
    rng = np.random.default_rng(1)
    lib = linearham_amd.load_library()
    fam = C.c_void_p(flat["family"])
    # the samples' site rates: K0's discrete-Gamma means, through the C ABI
    rates = np.zeros((n, R))
    _ll = np.zeros(n)

    from linearham_amd.capi import _EvalOutputs, c_f64p, c_i32p
    outs = _EvalOutputs()
    outs.rates = rates.ctypes.data_as(c_f64p)
    lib.check(lib.lib.lh_eval_batch(fam, n, T, depth, np.ascontiguousarray(flat["ops"]).ctypes.data_as(c_i32p),
                                    np.ascontiguousarray(flat["brlen"]).ctypes.data_as(c_f64p),
                                    flat["er"].ctypes.data_as(c_f64p), flat["pi"].ctypes.data_as(c_f64p),
                                    flat["alpha"].ctypes.data_as(c_f64p), R, _ll.ctypes.data_as(c_f64p),
                                    C.byref(outs)))
    naive = rng.integers(0, 4, size=(n, L)).astype(np.uint8)
    d = {k: torch.from_numpy(np.ascontiguousarray(flat[k])).to(dev) for k in ("ops", "brlen", "er", "pi")}
    d_rates, d_naive = torch.from_numpy(rates).to(dev), torch.from_numpy(naive).to(dev)
    anc = torch.zeros((n, T - 2, L), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(seed):
        lib.check(lib.lib.lh_asr_batch_device(fam, n, T, depth, d["ops"].data_ptr(), d["brlen"].data_ptr(),
                                              d["er"].data_ptr(), d["pi"].data_ptr(), d_rates.data_ptr(), R,
                                              d_naive.data_ptr(), seed, 0, anc.data_ptr(), None, C.c_void_p(stream)))
    for w in range(args.warmup):
        step(w)
    torch.cuda.synchronize()
    lib.check(lib.lib.lh_profile_enable(fam, 1))
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(100 + s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, k = C.c_double(), C.c_int64()
    lib.check(lib.lib.lh_asr_profile_read(fam, C.byref(ms), C.byref(k)))
    lib.check(lib.lib.lh_profile_enable(fam, 0))
    launches = max(k.value, 1)
    per_launch = n * args.steps / launches
    k3_ms = ms.value / launches
    bytes_per_sample = (T - 2) * L * (32 + 32 + 1)
    achieved = bytes_per_sample * per_launch / (k3_ms * 1e-3) / 1e9
    a = anc.cpu().numpy()
    # CPU baseline: the numpy restatement of the R step (oracle/asr_oracle.py) on a few samples of the same
    # batch, one core (the R script runs one tree per core); its draws must equal the GPU's
    from oracle import asr_oracle as ao
    from oracle import linearham_oracle as orc
    from linearham_amd import host as _host
    rows = sf.read_trees_tsv(os.path.join(fam_dir, "trees.tsv"))
    o = orc.PhyloHMM(os.path.join(fam_dir, "cluster.yaml"), 0, os.path.join(fam_dir, "hmm_params"), 0)
    labels = list(o.xmsa_labels)
    n_cpu = 4 if args.preset != "config4" else 1
    t_cpu = time.perf_counter()
    mism = 0
    last_seed = 100 + args.steps - 1
    for i in range(n_cpu):
        r = rows[i % len(rows)]
        children, root, brlen = _host.newick_arrays(r["tree"], labels)
        _, anc_ref, _ = ao.asr_sample(children, root, brlen, T, o.msa, naive[i], r["er"], np.asarray(r["pi"]),
                                      rates[i], last_seed, i)
        mism += int((anc_ref != a[i]).sum())
    t_cpu = time.perf_counter() - t_cpu
    if mism:
        raise SystemExit("parity failure: %d sampled states differ from the CPU oracle" % mism)
    out = {"metric": "ancestral-sequence samples/sec (per-site rate draw + joint inner-state draw)",
           "value": n * args.steps / dt, "unit": "tree samples/s", "ms_per_step": dt / args.steps * 1e3,
           "config": {"workload": args.preset, "batch": n, "n_tips": T, "n_sites": L, "R": R},
           "kernel_ms_per_launch": {"asr_K3": k3_ms, "samples_per_launch": per_launch},
           "roofline": {"bound": "hbm", "kernel": "asr_kernel (K3)", "achieved": achieved, "peak": 8000.0,
                        "unit": "GB/s", "frac": achieved / 8000.0, "algorithmic_bytes_per_sample": bytes_per_sample},
           "cpu_baseline": {"value": n_cpu / t_cpu, "unit": "tree samples/s", "cores": 1, "kind": "port",
                            "sample": "%d samples of the same batch through oracle/asr_oracle.py (numpy); sampled "
                                      "states identical to the GPU's" % n_cpu},
           "state_histogram": [int(x) for x in np.bincount(a.ravel(), minlength=4)[:4]]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

"""Randomised sweep of RunPipeline's device side (K0-K2 + the naive-sequence state draws K4; builder-run, not part of the pytest
suite): random small families -- igh / igk / igl, with and without N inside alignment columns, 1-8 alleles per segment -- through
the C++ host (host.PhyloHMM.run_pipeline, device sampler), every row's NaiveSequence and LogLikelihood against
oracle/linearham_oracle.py (same std::mt19937 stream; src/HMM.cpp:358-431, src/PhyloHMM.cpp RunPipeline).
usage (GPU box, repo root): python tests/dev_tools/random_sweep_pipeline.py [first_seed] [n_seeds] [--many] [--indels]"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from linearham_amd import host  # noqa: E402
from oracle import linearham_oracle as orc  # noqa: E402
from tools import synth_family as sf  # noqa: E402

many = "--many" in sys.argv     # 60-320 V / 10-80 D / 4-40 J alleles (K4's loops over more than 64 genes), three rows per family
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
first, n = (int(argv[0]) if len(argv) > 0 else 12000), (int(argv[1]) if len(argv) > 1 else 100)
bad = rows_checked = overflow_rows = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    locus = ["igh", "igh", "igk", "igl"][int(rng.integers(4))]
    kw = dict(locus=locus, seed=seed, n_samples=5, n_leaves=int(rng.integers(3, 40)), n_v=int(rng.integers(1, 9)),
              n_j=int(rng.integers(1, 6)), ragged=int(rng.choice([0, 0, 4, 10])), ambiguous=float(rng.choice([0.0, 0.0, 0.01, 0.05])),
              divergence=float(rng.choice([0.0, 0.05, 0.3])))
    if locus == "igh":
        kw["n_d"] = int(rng.integers(1, 6))
    if many:
        kw.update(n_samples=3, n_leaves=int(rng.integers(3, 12)), n_v=int(rng.integers(60, 320)), n_j=int(rng.integers(4, 40)))
        if locus == "igh":
            kw["n_d"] = int(rng.integers(10, 80))
    R = int(rng.choice([1, 3, 4]))
    rng_seed = int(rng.integers(0, 1000))
    if "--indels" in sys.argv:      # some sequences flagged has_shm_indels: read from indel_reversed_seqs (src/HMM.cpp:74-79)
        kw["shm_indels"] = int(rng.integers(1, 4))
    out = tempfile.mkdtemp(prefix="lh_sweepp_")
    try:
        sf.generate(sf.Spec.small(**kw), out)
        yaml_path, pdir, tsv = (os.path.join(out, x) for x in ("cluster.yaml", "hmm_params", "trees.tsv"))
        res = os.path.join(out, "lh.tsv")
        try:
            host.PhyloHMM(yaml_path, 0, pdir, rng_seed).run_pipeline(tsv, res, R)
        except RuntimeError as e:
            bad += 1
            print("seed", seed, locus, "R", R, "engine seed", rng_seed, {k: kw[k] for k in ("n_leaves", "n_v", "n_j", "ragged", "ambiguous", "divergence")},
                  kw.get("n_d"), ": RunPipeline raised:", str(e)[:200], flush=True)
            continue
        lines = [ln.rstrip("\n").split("\t") for ln in open(res)]
        c_seq, c_ll = lines[0].index("NaiveSequence"), lines[0].index("LHLogLikelihood") if "LHLogLikelihood" in lines[0] else None
        o = orc.PhyloHMM(yaml_path, 0, pdir, rng_seed)
        wrong = 0
        for r, got in zip(sf.read_trees_tsv(tsv), lines[1:]):
            o.initialize_phylo_parameters(r["tree"], r["er"], r["pi"], r["alpha"], R, is_path=False)
            o.initialize_phylo_emission()
            ll = o.log_likelihood()
            seq = o.sample_naive_sequence()
            rows_checked += 1
            if not np.isfinite(ll):   # the reference's 2^(256 d) equalisation overflow: non-finite on both sides;
                overflow_rows += 1    # what is drawn from inf / NaN weights follows each side's own inf / NaN pattern
                if c_ll is not None and np.isfinite(float(got[c_ll])):
                    wrong += 1
                continue
            seq_bad = got[c_seq] != seq
            ll_bad = c_ll is not None and abs(float(got[c_ll]) - ll) > 1e-5 * abs(ll)   # (the column is written with six significant digits, as the reference writes it)
            if seq_bad or ll_bad:
                wrong += 1
                if wrong == 1:
                    print("seed", seed, "first differing row:", "sequence" if seq_bad else "", "log-likelihood %s vs %r" % (got[c_ll], ll) if ll_bad else "",
                          flush=True)
        if wrong:
            bad += 1
            print("seed", seed, locus, "R", R, ":", wrong, "rows differ", flush=True)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    if (seed - first + 1) % (5 if many else 25) == 0:
        print("... %d seeds done" % (seed - first + 1), flush=True)
print("pipeline sweep of %d seeds from %d: %d families with differing rows; %d rows checked, %d of them reference overflow rows (non-finite on both sides)"
      % (n, first, bad, rows_checked, overflow_rows), flush=True)
sys.exit(1 if bad else 0)

"""Randomised sweep of the ancestral-sequence sampling step (K3 on K1's unmixed planes; builder-run, not part of the pytest
suite): random small families -- with and without N inside alignment columns, ladder-like and balanced trees, R in {1, 3, 4,
8} -- each tree sample's draws against oracle/asr_oracle.py (tests/test_gpu_asr._run; that oracle's parity is UNPINNED: the
reference holds no fixture for the step).  usage (GPU box, repo root): python tests/dev_tools/random_sweep_asr.py [first_seed] [n_seeds] [--large]"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import linearham_amd  # noqa: E402
import tests.test_gpu_asr as ta  # noqa: E402
from oracle import linearham_oracle as orc  # noqa: E402
from tools import synth_family as sf  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
first, n = (int(argv[0]) if len(argv) > 0 else 9000), (int(argv[1]) if len(argv) > 1 else 100)
lib = linearham_amd.load_library()
total_mism = total_sites = bad = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    locus = ["igh", "igh", "igk", "igl"][int(rng.integers(4))]
    kw = dict(locus=locus, seed=seed, n_samples=2, n_leaves=int(rng.integers(3, 70)), n_v=int(rng.integers(1, 5)),
              n_j=int(rng.integers(1, 4)), ragged=int(rng.choice([0, 0, 4, 10])), ambiguous=float(rng.choice([0.0, 0.0, 0.01, 0.05])),
              tree_shape=str(rng.choice(["stepwise", "stepwise", "balanced"])), n_nni=int(rng.integers(0, 4)))
    if locus == "igh":
        kw["n_d"] = int(rng.integers(1, 4))
    if "--large" in sys.argv:      # 100-400 leaves: K1's unfused kernels on deep / wide trees, K3's larger LDS tables
        kw["n_leaves"] = int(rng.integers(100, 400))
    out = tempfile.mkdtemp(prefix="lh_sweepa_")
    try:
        sf.generate(sf.Spec.small(**kw), out)
        h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
        rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
        R = int(rng.choice([1, 3, 4, 8]))
        mism, sites, _, _ = ta._run(lib, h, rows, R, seed=seed, first_sample=int(rng.integers(0, 1000)), rng=rng)
        total_mism += mism
        total_sites += sites
        if mism > max(1, sites // 200000):
            bad += 1
            print("seed", seed, "R", R, "leaves", kw["n_leaves"], ":", mism, "of", sites, "site draws differ", flush=True)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    if (seed - first + 1) % 25 == 0:
        print("... %d seeds done" % (seed - first + 1), flush=True)
print("asr sweep of %d seeds from %d: %d families beyond the bound; %d of %d site draws differ in all"
      % (n, first, bad, total_mism, total_sites), flush=True)
sys.exit(1 if bad else 0)

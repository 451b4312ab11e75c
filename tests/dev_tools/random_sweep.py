"""One-off randomised parity sweep (not part of the pytest suite): many seeds of
tests/test_gpu_parity.py::test_random_small_families plus the extended-range mode on the same families.
usage (GPU box, repo root): python tests/dev_tools/random_sweep.py [first_seed] [n_seeds]"""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import linearham_amd  # noqa: E402
import tests.test_gpu_parity as t  # noqa: E402
from oracle import linearham_oracle as orc  # noqa: E402
from tools import synth_family as sf  # noqa: E402

first, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
lib = linearham_amd.load_library()
bad = 0
for seed in range(first, first + n):
    d = pathlib.Path(tempfile.mkdtemp(prefix="lh_sweep_"))
    try:
        t.test_random_small_families(lib, d, seed)
        # the same family in the extended-range mode: same log-likelihood wherever the reference is finite
        out = str(d / "fam")
        h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
        rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
        R = int(np.random.default_rng(seed).choice([1, 3, 4]))
        _, ll, _, ref = t.run_family(lib, h, rows, R, extended=True)
        for i, r in enumerate(ref):
            if np.isfinite(r["loglik"]):
                assert abs(ll[i] - r["loglik"]) <= 1e-10 * abs(r["loglik"]), (seed, i, ll[i], r["loglik"])
            else:
                assert np.isfinite(ll[i]) or np.isneginf(ll[i]), (seed, i, ll[i])
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:300], flush=True)
print("sweep of %d seeds from %d: %d failures" % (n, first, bad), flush=True)
sys.exit(1 if bad else 0)

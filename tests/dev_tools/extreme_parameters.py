"""Extreme substitution-model parameters through the C ABI against the oracle (builder-run): one small family, tree samples whose GTR
rates span six decades, whose base frequencies come from Dirichlet(0.2) (components down to 1e-6), alpha log-uniform in [0.005, 1000],
branch lengths scaled by 1e-2 .. 1e2 -- K0a's eigen-decomposition and discrete-Gamma solver, K1's P-matrices and rescaling away
from the generator's comfortable ranges.  What to expect (profiles/r04_extreme_parameters.txt): rates to 1e-11 everywhere; with base
frequencies of 1e-6 the two CPU restatements themselves differ by up to 5e-9 on the log-likelihood (the eigen-decomposition of the
symmetrised rate matrix: LAPACK's eigh in numpy, cyclic Jacobi in C), and the kernels -- Jacobi too -- side with the C one to 1e-15.
usage (GPU box, repo root): python tests/dev_tools/extreme_parameters.py [seed] [n_rows]"""
import os
import re
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import linearham_amd  # noqa: E402
import tests.test_gpu_parity as t  # noqa: E402
from oracle import linearham_oracle as orc  # noqa: E402
from tools import synth_family as sf  # noqa: E402

seed, n_rows = (int(sys.argv[1]) if len(sys.argv) > 1 else 3), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
rng = np.random.default_rng(seed)
lib = linearham_amd.load_library()
out = tempfile.mkdtemp(prefix="lh_extreme_")
worst = {"loglik": 0.0, "rates": 0.0, "emission": 0.0}
bad = skipped = 0
try:
    sf.generate(sf.Spec.small(n_leaves=14, n_samples=8, seed=seed, ragged=4, ambiguous=0.02), out)
    h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    base = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    for i in range(n_rows):
        r = dict(base[i % len(base)])
        scale = 10.0 ** rng.uniform(-2, 2)
        r["tree"] = re.sub(r":([0-9.eE+-]+)", lambda m: ":%.12g" % max(float(m.group(1)) * scale, 1e-6), r["tree"])
        r["er"] = (10.0 ** rng.uniform(-3, 3, 6)).tolist()
        pi = rng.dirichlet(np.ones(4) * 0.2)
        pi = np.maximum(pi, 1e-6)
        r["pi"] = (pi / pi.sum()).tolist()
        r["alpha"] = float(10.0 ** rng.uniform(np.log10(0.005), 3))
        R = int(rng.choice([1, 2, 4, 8]))
        desc, ll, res, ref = t.run_family(lib, h, [r], R)
        if not np.isfinite(ref[0]["loglik"]):
            skipped += 1
            assert not np.isfinite(ll[0])
            continue
        worst["loglik"] = max(worst["loglik"], abs(ll[0] - ref[0]["loglik"]) / abs(ref[0]["loglik"]))
        worst["rates"] = max(worst["rates"], float(np.max(np.abs(res["rates"][0] - ref[0]["rates"]) / np.maximum(ref[0]["rates"], 1e-300))))
        e, g = np.asarray(ref[0]["xmsa_emission"]), np.asarray(res["xmsa_emission"][0])
        m = e > 1e-290
        if m.any():
            worst["emission"] = max(worst["emission"], float(np.max(np.abs(g[m] - e[m]) / e[m])))
        try:
            t.compare(h, desc, ll, res, ref)
        except AssertionError as err:
            bad += 1
            print("row", i, "R", R, "alpha %.4g scale %.3g pi min %.2g er %s" % (r["alpha"], scale, min(r["pi"]), np.round(r["er"], 4).tolist()),
                  "FAILED", " ".join(str(err).split())[:260], flush=True)
        if (i + 1) % 50 == 0:
            print("... %d rows done" % (i + 1), flush=True)
finally:
    shutil.rmtree(out, ignore_errors=True)
print("extreme parameters, %d rows: %d outside the suite's bounds, %d reference overflow rows; largest relative deviations: log-likelihood %.1e, rates %.1e, emissions %.1e"
      % (n_rows, bad, skipped, worst["loglik"], worst["rates"], worst["emission"]), flush=True)
sys.exit(1 if bad else 0)

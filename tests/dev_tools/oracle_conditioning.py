"""(History: this tool is what showed, at the end of round 4, that the "conditioning" it was written to measure was the
oracles' own -- P-matrices formed with exp() instead of libpll's expm1 form; with that form the two restatements agree to
1e-15 / 1e-11 on every former outlier.  It stays as the check that they do.)
How well conditioned is the reference algorithm on a sweep family?  Rebuilds the family tests/dev_tools/random_sweep_forms.py
draws for a seed and compares the two CPU restatements (numpy and C: the same operations in another summation order) on its
forward arrays.  A seed on which they differ by more than the suite's 1e-8 is one where that bound says nothing about a third
implementation.  CPU only.  usage: python tests/dev_tools/oracle_conditioning.py [--wide] seed [seed ...]"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import linearham_oracle as orc  # noqa: E402
from oracle import oracle_c  # noqa: E402
from tests import desc_builder as db  # noqa: E402
from tools import synth_family as sf  # noqa: E402


WIDE = "--wide" in sys.argv


def spec_of(seed):
    """the family of random_sweep_forms.py for this seed (same draws in the same order) and its R"""
    rng = np.random.default_rng(seed)
    locus = ["igh", "igh", "igk", "igl"][int(rng.integers(4))]
    kw = dict(locus=locus, seed=seed, n_samples=2, n_nni=int(rng.integers(0, 4)),
              ragged=int(rng.choice([0, 0, 4, 10])), ambiguous=float(rng.choice([0.0, 0.0, 0.01, 0.05])),
              tree_shape=str(rng.choice(["stepwise", "stepwise", "balanced"])))
    if WIDE:
        kw.update(n_leaves=int(rng.integers(40, 110)), n_sites=600, n_v=int(rng.integers(4, 24)), n_d=int(rng.integers(1, 6)),
                  n_j=int(rng.integers(1, 4)), brlen_mean=float(rng.choice([0.01, 0.02, 0.03])))
        if locus != "igh":
            kw.pop("n_d")
        spec = sf.Spec(**kw)
    elif rng.random() < 0.35:
        kw.update(n_leaves=int(rng.integers(20, 70)), n_sites=240, len_v=150, len_d=(8, 20), len_j=(30, 45),
                  n_v=int(rng.integers(2, 8)), n_j=int(rng.integers(1, 4)), v_ancestors=2, d_ancestors=2, j_ancestors=2,
                  divergence=0.1, brlen_mean=float(rng.choice([0.01, 0.03])))
        if locus == "igh":
            kw["n_d"] = int(rng.integers(1, 4))
        spec = sf.Spec(**kw)
    else:
        kw.update(n_leaves=int(rng.integers(3, 70)), n_v=int(rng.integers(1, 9)), n_j=int(rng.integers(1, 6)),
                  divergence=float(rng.choice([0.0, 0.05, 0.3])))
        if locus == "igh":
            kw["n_d"] = int(rng.integers(1, 6))
        spec = sf.Spec.small(**kw)
    return spec, rng


for seed in (int(a) for a in sys.argv[1:] if not a.startswith("--")):
    spec, rng = spec_of(seed)
    out = tempfile.mkdtemp(prefix="lh_cond_")
    try:
        sf.generate(spec, out)
        h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
        rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
        R = int(rng.choice([2, 3, 4, 5, 8])) if WIDE else int(rng.choice([1, 3, 4]))
        fam = oracle_c.COracleFamily(h, R)
        trees = [db.tree_arrays(orc.parse_newick(r["tree"]), h.xmsa_labels) for r in rows]
        got = fam.eval_forward(trees, [r["er"] for r in rows], [r["pi"] for r in rows], [r["alpha"] for r in rows])
        for i, (r, g) in enumerate(zip(rows, got)):
            h.initialize_phylo_parameters(r["tree"], r["er"], r["pi"], r["alpha"], R, is_path=False)
            h.initialize_phylo_emission()
            ref = h.log_likelihood()
            worst, where = 0.0, ""
            for k, v in g.items():
                if k == "loglik" or "scaler" in k:
                    continue
                want, v = np.asarray(getattr(h, k), dtype=float), np.asarray(v, dtype=float)
                m = want != 0
                if m.any():
                    d = float(np.max(np.abs(v[m] - want[m]) / np.abs(want[m])))
                    if d > worst:
                        worst, where = d, k
            print("seed %d sample %d (R = %d, alpha %.4g): numpy %.12f  C %.12f: log-likelihood %.1e relative, forward arrays up to %.1e (%s)"
                  % (seed, i, R, r["alpha"], ref, g["loglik"], abs(g["loglik"] - ref) / abs(ref), worst, where), flush=True)
    finally:
        shutil.rmtree(out, ignore_errors=True)

"""Randomised parity sweep over K1's kernel forms (builder-run, not part of the pytest suite): families drawn with N inside
alignment columns (ragged reads, ambiguous bases) or without, ladder-like or balanced trees, one or several waves per rate;
every one compared with the numpy oracle (tests/test_gpu_parity.compare), the form each reached (lh_family_prune_form)
tallied.  usage (GPU box, repo root): python tests/dev_tools/random_sweep_forms.py [first_seed] [n_seeds] [--wide | --many | --huge | --large | --long] [--ext]"""
import collections
import os
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

wide = "--wide" in sys.argv      # 600-site families of 40-110 leaves: three to five waves per rate, R in {2, 3, 4, 5, 8}
many = "--many" in sys.argv or "--huge" in sys.argv      # 3-10 leaves, 30-320 V / 1-140 D / 1-140 J alleles: K2b's multi-chunk junction kernels, K2a's gene slots
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
first, n = (int(argv[0]) if len(argv) > 0 else 5000), (int(argv[1]) if len(argv) > 1 else 100)
lib = linearham_amd.load_library()


def loose(h, desc, ll, res, ref):
    """compare() with an absolute floor of 1e-15 (relative to a vector's largest entry) on emissions / forward entries."""
    for i, r in enumerate(ref):
        assert abs(ll[i] - r["loglik"]) <= 1e-10 * abs(r["loglik"]), (i, ll[i], r["loglik"])
        np.testing.assert_allclose(res["xmsa_emission"][i], r["xmsa_emission"], rtol=1e-8, atol=1e-15)
        ex = t.expand_forward(h, desc, res["forward"][i], res["scaler_counts"][i])
        for k in ex:
            if "scaler" in k:
                assert ex[k] == r[k], (i, k)
            else:
                scale = float(np.max(np.abs(r[k]))) if np.size(r[k]) else 0.0
                np.testing.assert_allclose(ex[k], r[k], rtol=1e-8, atol=1e-15 * scale, err_msg="%d %s" % (i, k))


forms = collections.Counter()
bad = soft = skipped = 0
worst = {"loglik": 0.0, "emission": 0.0, "forward": 0.0}   # largest relative deviations from the oracle seen in the sweep


def deviations(h, desc, ll, res, ref):
    for i, r in enumerate(ref):
        worst["loglik"] = max(worst["loglik"], abs(ll[i] - r["loglik"]) / abs(r["loglik"]))
        e, g = np.asarray(r["xmsa_emission"]), np.asarray(res["xmsa_emission"][i])
        m = e != 0
        if m.any():
            worst["emission"] = max(worst["emission"], float(np.max(np.abs(g[m] - e[m]) / e[m])))
        ex = t.expand_forward(h, desc, res["forward"][i], res["scaler_counts"][i])
        for k in ex:
            if k.endswith("_forward"):
                w, v = np.asarray(r[k], dtype=float), np.asarray(ex[k], dtype=float)
                m = w != 0
                if m.any():
                    worst["forward"] = max(worst["forward"], float(np.max(np.abs(v[m] - w[m]) / np.abs(w[m]))))
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    locus = ["igh", "igh", "igk", "igl"][int(rng.integers(4))]
    kw = dict(locus=locus, seed=seed, n_samples=2, n_nni=int(rng.integers(0, 4)),
              ragged=int(rng.choice([0, 0, 4, 10])), ambiguous=float(rng.choice([0.0, 0.0, 0.01, 0.05])),
              tree_shape=str(rng.choice(["stepwise", "stepwise", "balanced"])))
    if "--long" in sys.argv:         # 1500-3000 sites, 40-80 leaves: more than 1024 patterns -- several site tiles per (sample, rate) by shape
        kw.update(n_leaves=int(rng.integers(40, 80)), n_sites=int(rng.integers(1500, 3000)), len_v=int(rng.integers(1300, 1400)),
                  n_v=int(rng.integers(2, 6)), n_j=int(rng.integers(1, 3)), brlen_mean=float(rng.choice([0.02, 0.04])))
        if locus == "igh":
            kw["n_d"] = int(rng.integers(1, 3))
        spec = sf.Spec(**kw)
    elif "--large" in sys.argv:        # 120-400 leaves (tip tables beyond a workgroup's share of LDS: the segmented kernels by shape)
        kw.update(n_leaves=int(rng.integers(120, 400)), n_v=int(rng.integers(1, 9)), n_j=int(rng.integers(1, 6)))
        if locus == "igh":
            kw["n_d"] = int(rng.integers(1, 6))
        spec = sf.Spec.small(**kw)
    elif many:
        huge = "--huge" in sys.argv      # up to the ABI's limits: 1024 V, 256 D, 256 J alleles (sixteen / four register chunks)
        kw.update(n_leaves=int(rng.integers(3, 10)), n_v=int(rng.integers(500, 1025) if huge else rng.integers(30, 320)),
                  n_j=int(rng.integers(100, 257) if huge else rng.integers(1, 140)), divergence=float(rng.choice([0.0, 0.05, 0.3])))
        if locus == "igh":
            kw["n_d"] = int(rng.integers(100, 257) if huge else rng.integers(1, 140))
        spec = sf.Spec.small(**kw)
    elif wide:
        kw.update(n_leaves=int(rng.integers(40, 110)), n_sites=600, n_v=int(rng.integers(4, 24)), n_d=int(rng.integers(1, 6)),
                  n_j=int(rng.integers(1, 4)), brlen_mean=float(rng.choice([0.01, 0.02, 0.03])))
        if locus != "igh":
            kw.pop("n_d")
        spec = sf.Spec(**kw)
    elif rng.random() < 0.35:      # several waves per rate: 240 sites, longer V
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
    out = tempfile.mkdtemp(prefix="lh_sweepf_")
    try:
        sf.generate(spec, out)
        h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
        rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
        R = int(rng.choice([2, 3, 4, 5, 8])) if wide else int(rng.choice([1, 3, 4]))
        desc, ll, res, ref = t.run_family(lib, h, rows, R)
        forms[t.LAST_RUN["form"]] += 1
        worst["patterns"] = max(worst.get("patterns", 0), t.LAST_RUN["n_patterns"])
        if not all(np.isfinite(r["loglik"]) for r in ref):   # the reference's own overflow rows: same mask on both sides
            assert [bool(np.isfinite(x)) for x in ll] == [bool(np.isfinite(r["loglik"])) for r in ref], seed
            skipped += 1
            continue
        deviations(h, desc, ll, res, ref)
        if "--ext" in sys.argv:      # the same family in the extended-range mode: the same log-likelihood wherever the reference is finite
            _, ll_x, _, _ = t.run_family(lib, h, rows, R, extended=True)
            for i, r in enumerate(ref):
                if np.isfinite(r["loglik"]) and not abs(ll_x[i] - r["loglik"]) <= 1e-12 * abs(r["loglik"]):
                    bad += 1
                    print("seed", seed, "extended-range mode FAILED", i, ll_x[i], r["loglik"], flush=True)
        try:
            t.compare(h, desc, ll, res, ref)
        except AssertionError as e:
            try:
                loose(h, desc, ll, res, ref)
                soft += 1
                print("seed", seed, t.LAST_RUN["form"], "beyond 1e-8 relative on tiny entries only:", " ".join(str(e).split())[:160], flush=True)
            except AssertionError as e2:
                bad += 1
                print("seed", seed, t.LAST_RUN["form"], "FAILED", " ".join(str(e2).split())[:300], flush=True)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    if (seed - first + 1) % (5 if wide or many or '--large' in sys.argv or '--long' in sys.argv else 50) == 0:
        print("... %d seeds done" % (seed - first + 1), flush=True)
print("sweep of %d seeds from %d: %d failures, %d beyond the relative tolerance on tiny entries only, %d with reference overflow rows"
      % (n, first, bad, soft, skipped), flush=True)
print("largest relative deviations from the oracle: log-likelihood %.1e, xMSA emissions %.1e, forward entries %.1e"
      % (worst["loglik"], worst["emission"], worst["forward"]), flush=True)
print("most site patterns in a family:", worst.get("patterns", 0), flush=True)
print("forms reached:", dict(sorted(forms.items())), flush=True)
sys.exit(1 if bad else 0)

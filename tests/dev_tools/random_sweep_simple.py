"""Randomised sweep of the SimpleHMM path (lh_forward_batch: K2 on emissions computed on the host; src/SimpleHMM.cpp, BASELINE.json
configs[0]; builder-run): random families -- igh / igk / igl, 1-8 or (--many) 30-320 V / 1-140 D / 1-140 J alleles, ragged reads and
ambiguous bases in half of them -- through the C++ host's SimpleHMM: log-likelihood, forward arrays, ScaleMatrix counts and the
sampled naive sequence against oracle/linearham_oracle.py's SimpleHMM (same std::mt19937 stream).
usage (GPU box, repo root): python tests/dev_tools/random_sweep_simple.py [first_seed] [n_seeds] [--many]"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from linearham_amd import host  # noqa: E402
from oracle import linearham_oracle as orc  # noqa: E402
from tests import test_host_gpu as thg  # noqa: E402
from tools import synth_family as sf  # noqa: E402

many = "--many" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
first, n = (int(argv[0]) if len(argv) > 0 else 16000), (int(argv[1]) if len(argv) > 1 else 100)
bad = refused = overflow = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    locus = ["igh", "igh", "igk", "igl"][int(rng.integers(4))]
    kw = dict(locus=locus, seed=seed, n_samples=1, n_leaves=int(rng.integers(2, 30)), n_v=int(rng.integers(1, 9)),
              n_j=int(rng.integers(1, 6)), ragged=int(rng.choice([0, 0, 4, 10])), ambiguous=float(rng.choice([0.0, 0.0, 0.01, 0.05])),
              divergence=float(rng.choice([0.0, 0.05, 0.3])))
    if locus == "igh":
        kw["n_d"] = int(rng.integers(1, 6))
    if many:
        kw.update(n_v=int(rng.integers(30, 320)), n_j=int(rng.integers(1, 140)))
        if locus == "igh":
            kw["n_d"] = int(rng.integers(1, 140))
    engine_seed = int(rng.integers(0, 1000))
    out = tempfile.mkdtemp(prefix="lh_sweeps_")
    try:
        sf.generate(sf.Spec.small(**kw), out)
        yaml_path, pdir = os.path.join(out, "cluster.yaml"), os.path.join(out, "hmm_params")
        o = orc.SimpleHMM(yaml_path, 0, pdir, engine_seed)
        try:
            h = host.SimpleHMM(yaml_path, 0, pdir, engine_seed)
            ll, ref = h.log_likelihood(), o.log_likelihood()
        except RuntimeError as e:
            if "too large for the forward kernels' LDS working set" in str(e):   # (documented limit of lh_family_create)
                refused += 1
                continue
            raise
        if not np.isfinite(ref):      # the reference's 2^(256 d) equalisation overflow: non-finite on both sides
            overflow += 1
            if np.isfinite(ll):
                bad += 1
                print("seed", seed, "finite", ll, "where the reference overflows", ref, flush=True)
            continue
        try:
            assert abs(ll - ref) <= 1e-12 * abs(ref), (ll, ref)
            d = h.dump(2)
            for k in thg.FWD_KEYS:
                if hasattr(o, k) and getattr(o, k) is not None and k in d:
                    np.testing.assert_allclose(np.asarray(d[k], dtype=float), getattr(o, k), rtol=1e-9, atol=0, err_msg=k)
            for k in thg.CNT_KEYS:
                if hasattr(o, k) and k in d:
                    assert d[k] == getattr(o, k), k
            assert h.sample_naive_sequence() == o.sample_naive_sequence(), "sampled naive sequence"
        except AssertionError as e:
            bad += 1
            print("seed", seed, locus, {k: kw.get(k) for k in ("n_v", "n_d", "n_j", "ragged", "ambiguous", "divergence")}, "FAILED",
                  " ".join(str(e).split())[:300], flush=True)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    if (seed - first + 1) % (10 if many else 50) == 0:
        print("... %d seeds done" % (seed - first + 1), flush=True)
print("SimpleHMM sweep of %d seeds from %d: %d failures; %d families refused by lh_family_create (too many columns for the LDS working set)"
      % (n, first, bad, refused), flush=True)
print("(%d families on which the reference overflows: non-finite on both sides)" % overflow, flush=True)
sys.exit(1 if bad else 0)

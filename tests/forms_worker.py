"""Helper of tests/test_gpu_forms.py (its own process: the K1 form hooks are environment variables read once per
process).  For every family named on the command line: generate it, evaluate it through the C ABI, compare log-likelihood,
rates, emissions, forward arrays and ScaleMatrix counts with the numpy oracle (tests/test_gpu_parity.py: run_family +
compare), and report which pruning-kernel form ran (lh_family_prune_form).  Prints one JSON line {family: {...}}.
`--oracle-only` runs the CPU side alone (no GPU): used to check that the families evaluate finitely."""
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def specs():
    from tools import synth_family as sf
    mid = dict(n_sites=400, n_v=24, n_d=6, n_j=4, n_samples=3)
    return {
        # shapes without N inside alignment columns
        "small_igh": sf.Spec.small(n_samples=4, seed=40),                                   # one one-site wave per rate
        "mid60x400": sf.Spec(n_leaves=60, seed=41, **mid),                                  # > 128 patterns: two-site waves
        "balanced64": sf.Spec.small(n_leaves=64, n_samples=3, seed=47, tree_shape="balanced", n_nni=0),   # stack depth >= 5
        "wide100x600": sf.Spec(n_leaves=100, n_sites=600, n_v=24, n_d=6, n_j=4, n_samples=3, seed=48, brlen_mean=0.02),
        "wide100x600_r8": sf.Spec(n_leaves=100, n_sites=600, n_v=24, n_d=6, n_j=4, n_samples=3, seed=48, brlen_mean=0.02),
        "wide100x600_r3": sf.Spec(n_leaves=100, n_sites=600, n_v=24, n_d=6, n_j=4, n_samples=3, seed=48, brlen_mean=0.02),
        # ragged reads (N padding of unequal extent at both ends) and scattered ambiguous bases: libpll's N = 1111
        # (src/HMM.cpp:69-83, src/PhyloHMM.cpp:229-235,368-370)
        "mixed_small": sf.Spec.small(n_leaves=20, n_samples=3, seed=42, ragged=6, ambiguous=0.02),
        "mixed_igk": sf.Spec.small(locus="igk", n_leaves=12, n_samples=3, seed=43, ragged=5, ambiguous=0.02),
        "mixed_120": sf.Spec.small(n_leaves=120, n_samples=3, seed=44, ragged=6, ambiguous=0.01),
        "mixed_60x400": sf.Spec(n_leaves=60, seed=45, ragged=30, ambiguous=0.01, **mid),
        "mixed_500": sf.Spec(n_leaves=500, n_sites=600, n_v=24, n_d=6, n_j=4, n_samples=2, seed=46, ragged=30, ambiguous=0.01),
        "mixed_balanced64": sf.Spec.small(n_leaves=64, n_samples=3, seed=49, tree_shape="balanced", n_nni=0, ragged=6,
                                          ambiguous=0.01),
    }


# Bounds other than compare()'s (none since the oracles form their P-matrices with expm1: mixed_500's first tree sample used
# to need 2e-8 / 1e-7 against the exp() form's rounding noise on 1e-6 branches).
TOLERANCE = {}
NUM_RATES = {"wide100x600_r8": 8, "wide100x600_r3": 3}     # (4 everywhere else)


def main(argv):
    import numpy as np
    from oracle import linearham_oracle as orc
    from tools import synth_family as sf
    oracle_only = "--oracle-only" in argv
    names = [a for a in argv if not a.startswith("--")]
    all_specs = specs()
    report = {}
    if not oracle_only:
        import linearham_amd
        from tests import test_gpu_parity as tp
        hip = linearham_amd.load_library()
        assert hip.device_count() >= 1, "no HIP device visible"
    for name in names:
        out = tempfile.mkdtemp(prefix="lh_forms_")
        try:
            sf.generate(all_specs[name], out)
            h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
            rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
            n_n = (h.msa == 4).sum(axis=0)
            info = {"mixed_columns": int(((n_n > 0) & (n_n < h.msa.shape[0])).sum())}
            if oracle_only:
                lls = []
                for r in rows:
                    h.initialize_phylo_parameters(r["tree"], r["er"], r["pi"], r["alpha"], NUM_RATES.get(name, 4), is_path=False)
                    h.initialize_phylo_emission()
                    lls.append(float(h.log_likelihood()))
                info["oracle"] = lls
            else:
                desc, ll, res, ref = tp.run_family(hip, h, rows, NUM_RATES.get(name, 4))
                assert all(np.isfinite(r["loglik"]) for r in ref), "the family should evaluate finitely in the reference"
                if "--report" in argv:   # development: the largest deviations instead of the assertions
                    dev = {"loglik": 0.0, "emission": 0.0}
                    for i, r in enumerate(ref):
                        dev["loglik"] = max(dev["loglik"], abs(ll[i] - r["loglik"]) / abs(r["loglik"]))
                        e = np.abs(res["xmsa_emission"][i] - r["xmsa_emission"]) / np.abs(r["xmsa_emission"])
                        dev["emission"] = max(dev["emission"], float(np.nanmax(e)))
                        ex = tp.expand_forward(h, desc, res["forward"][i], res["scaler_counts"][i])
                        for k in ex:
                            if k.endswith("_forward"):
                                a_, b_ = np.asarray(ex[k], float), np.asarray(r[k], float)
                                m = b_ != 0
                                dev[k] = max(dev.get(k, 0.0), float(np.max(np.abs(a_[m] - b_[m]) / np.abs(b_[m]))) if m.any() else 0.0)
                            else:
                                dev[k] = dev.get(k, True) and bool(np.array_equal(np.asarray(ex[k]), np.asarray(r[k])))
                    info["deviation"] = dev
                else:
                    tp.compare(h, desc, ll, res, ref, **TOLERANCE.get(name, {}))
                info.update(tp.LAST_RUN, loglik=[float(x) for x in ll])
            report[name] = info
        finally:
            shutil.rmtree(out, ignore_errors=True)
    print(json.dumps(report))


if __name__ == "__main__":
    main(sys.argv[1:])

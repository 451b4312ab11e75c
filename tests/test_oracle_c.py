"""The plain-C restatement (oracle/oracle_kernels.c, bench.py's cpu_baseline) against the numpy oracle."""
import os

import numpy as np
import pytest

from oracle import linearham_oracle as orc
from oracle import oracle_c
from tests import desc_builder as db

ER, PI = [1.0] * 6, [0.17, 0.19, 0.25, 0.39]


@pytest.mark.parametrize("case,params,R", [("phylo_hmm_input", "hmm_params", 4),
                                          ("phylo_hmm_input_extra", "hmm_params", 4),
                                          ("phylo_likelihood_hmm_input", "phylo_likelihood_hmm_params", 1)])
def test_c_oracle_on_reference_fixtures(goldens, data_dir, case, params, R):
    ll, h = orc.phylo_loglik(os.path.join(data_dir, case + ".yaml"), os.path.join(data_dir, params),
                             os.path.join(data_dir, "newton.tree"), ER, PI, 1.0, R)
    fam = oracle_c.COracleFamily(h, R)
    tree = db.tree_arrays(h.tree, h.xmsa_labels)
    got, em = fam.eval([tree], [ER], [PI], [1.0], want_em=True)
    assert abs(got[0] - ll) < 1e-12 * abs(ll)
    np.testing.assert_allclose(em[0], h.xmsa_emission, rtol=1e-12)
    from tests.helpers import catch_approx
    assert catch_approx(got[0], goldens["PhyloHMM:" + case]["vars"]["loglikelihood"])


@pytest.mark.parametrize("kw", [dict(n_leaves=40, n_samples=4, seed=11),
                                dict(n_leaves=20, n_samples=3, seed=42, ragged=6, ambiguous=0.02),    # N inside columns
                                dict(n_leaves=64, n_samples=2, seed=47, tree_shape="balanced", n_nni=0)],
                         ids=["medium", "mixed_n", "balanced"])
def test_c_oracle_on_synthetic_family(tmp_path, kw):
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec.small(**kw), out)
    h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    fam = oracle_c.COracleFamily(h, 4)
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    trees = [db.tree_arrays(orc.parse_newick(r["tree"]), h.xmsa_labels) for r in rows]
    got = fam.eval(trees, [r["er"] for r in rows], [r["pi"] for r in rows], [r["alpha"] for r in rows], n_threads=2)
    for r, g in zip(rows, got):
        h.initialize_phylo_parameters(r["tree"], r["er"], r["pi"], r["alpha"], 4, is_path=False)
        h.initialize_phylo_emission()
        ref = h.log_likelihood()
        assert abs(g - ref) < 1e-12 * abs(ref), (g, ref)


def test_c_oracle_forward_arrays_match_the_numpy_oracle(tmp_path):
    """oc_eval_batch_fwd (the arrays SampleNaiveSequence reads, src/HMM.cpp:326,1250,1333) against the numpy oracle's
    members of the same names: values 1e-10 (products of some 300 emissions), ScaleMatrix counts exactly, heavy and light chain."""
    from tools import synth_family as sf
    for kw in (dict(n_leaves=40, n_samples=3, seed=11), dict(locus="igk", n_samples=2, seed=5)):
        out = str(tmp_path / ("fam_%s" % kw.get("locus", "igh")))
        sf.generate(sf.Spec.small(**kw), out)
        h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
        fam = oracle_c.COracleFamily(h, 4)
        rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
        trees = [db.tree_arrays(orc.parse_newick(r["tree"]), h.xmsa_labels) for r in rows]
        got = fam.eval_forward(trees, [r["er"] for r in rows], [r["pi"] for r in rows], [r["alpha"] for r in rows])
        for r, g in zip(rows, got):
            h.initialize_phylo_parameters(r["tree"], r["er"], r["pi"], r["alpha"], 4, is_path=False)
            h.initialize_phylo_emission()
            ref = h.log_likelihood()
            assert abs(g["loglik"] - ref) < 1e-12 * abs(ref)
            for k, v in g.items():
                if k == "loglik":
                    continue
                want = getattr(h, k)
                if "scaler" in k:
                    assert np.array_equal(np.asarray(v), np.asarray(want)), k
                else:
                    np.testing.assert_allclose(v, want, rtol=1e-10, atol=0, err_msg=k)

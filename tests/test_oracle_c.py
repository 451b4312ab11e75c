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


def test_pmatrix_form_on_tiny_qt():
    """gtr_pmatrices forms P as libpll's core_pmatrix.c does (expm1 of the eigenvalues, identity added at the end).  On a 1e-6
    branch at the slowest rate of alpha = 0.05 (Q t r ~ 1e-19) its off-diagonal entries are Q_ij t r to rounding; the plain
    exp() form this oracle had until round 4 leaves rounding noise of size 1e-17 there -- two orders above the entries, of
    either sign.  On ordinary branches the two forms agree to 1e-15 absolute."""
    er, pi = [1.0, 2.5, 0.7, 1.3, 3.1, 0.9], np.array([0.17, 0.19, 0.25, 0.39])
    rates = orc.gamma_rates_mean(0.05, 4)
    assert rates[0] < 1e-11
    t = np.array([1e-6, 0.01, 0.3])
    P = orc.gtr_pmatrices(er, pi, rates, t)
    P_exp = orc.gtr_pmatrices(er, pi, rates, t, plain_exp=True)
    np.testing.assert_allclose(P.sum(axis=-1), 1.0, rtol=0, atol=4e-15)
    # the generator, as gtr_pmatrices builds it
    S = np.zeros((4, 4))
    S[np.triu_indices(4, 1)] = er
    S = S + S.T
    Q = S * pi[None, :]
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(axis=1))
    Q /= -np.sum(pi * np.diag(Q))
    off = ~np.eye(4, dtype=bool)
    tiny = P[0, 0][off]                                   # 1e-6 branch, slowest rate
    np.testing.assert_allclose(tiny, (Q * t[0] * rates[0])[off], rtol=1e-12)
    assert (tiny > 0).all()
    noise = np.abs(P_exp[0, 0][off] - tiny)
    assert noise.max() > 10 * tiny.max()                  # the old form: noise, not signal, at this size
    np.testing.assert_allclose(P[1:, 2:], P_exp[1:, 2:], rtol=0, atol=2e-15)

"""CPU tests of the ancestral-sequence oracle (oracle/asr_oracle.py; scripts/run_bootstrap_asr_ess.R:48-104):
the random-number generator against Random123's known-answer vectors, and the sampler against brute-force
enumeration of the joint posterior it is meant to draw from."""
import numpy as np

from oracle import asr_oracle as ao
from oracle import linearham_oracle as orc


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32-10."""
    hx = lambda x: [int(v) for v in x]
    assert hx(ao.philox4x32_10(0, 0, 0, 0, 0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert hx(ao.philox4x32_10(f, f, f, f, f, f)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert hx(ao.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    u = ao.uniform(7, 3, np.arange(100000), 5)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 5e-3


def test_draw_rule():
    w = np.array([[0.0, 2.0, 0.0, 2.0], [1.0, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0]])
    assert list(ao.draw(w, np.array([0.0, 0.0, 0.3]))) == [1, 0, 3]
    assert list(ao.draw(w, np.array([0.49999, 0.25, 0.9]))) == [1, 1, 3]
    assert list(ao.draw(w, np.array([0.5, 0.999, 0.0]))) == [3, 3, 3]


def _tree5():
    # tips 0 (naive), 1..4; inner 5 (root, naive's neighbour), 6, 7
    children = np.array([6, 7, 1, 2, 3, 4], dtype=np.int32)     # 5:(6,7) 6:(1,2) 7:(3,4)
    brlen = np.array([0.11, 0.2, 0.05, 0.3, 0.15, 0.0, 0.25, 0.07])
    return children, 5, brlen, 5


def test_sampler_draws_from_the_joint_posterior():
    children, root, brlen, T = _tree5()
    er = [1.3, 0.7, 2.1, 0.4, 1.1, 0.9]
    pi = np.array([0.17, 0.19, 0.25, 0.39])
    rates = [0.3, 1.7]
    N = 60000
    for column in ([2, 0, 2, 3, 4], [4, 1, 1, 1, 0]):      # naive G / N; a tip with N; a near-constant column
        msa = np.repeat(np.array(column[1:])[:, None], N, axis=1)
        naive = np.full(N, column[0])
        choice, anc, detail = ao.asr_sample(children, root, brlen, T, msa, naive, er, pi, rates, seed=11, sample_index=4)
        # rate weights: likelihood of the column on each scaled tree
        lk = np.exp(detail["loglik_per_rate"][:, 0])
        want = lk / lk.sum()
        got = np.bincount(choice, minlength=2) / N
        assert np.all(np.abs(got - want) < 4 * np.sqrt(want * (1 - want) / N) + 1e-9)
        # site likelihood = sum over the joint enumeration (cross-check of the pruning against brute force)
        for k, r in enumerate(rates):
            post = ao.exact_joint_posterior(children, root, brlen, T, column, er, pi, r)
            sel = choice == k
            n_k = int(sel.sum())
            idx = anc[0, sel].astype(int) * 16 + anc[1, sel].astype(int) * 4 + anc[2, sel].astype(int)
            emp = np.bincount(idx, minlength=64) / n_k
            exp = post.ravel()
            z = np.abs(emp - exp) / np.sqrt(np.maximum(exp * (1 - exp), 1e-12) / n_k)
            assert z[exp > 1e-4].max() < 4.5, (column, k, z.max())
            assert emp[exp < 1e-9].sum() == 0.0


def test_site_likelihoods_are_the_pinned_pruning_values(data_dir):
    """The per-rate column likelihoods of the ASR oracle are the pruning values that the reference's goldens pin
    (oracle/linearham_oracle.py::per_site_loglik), on the toy family's tree."""
    from tests import desc_builder as db
    tree = orc.parse_newick(open(data_dir + "/newton.tree").read())
    labels = ["naive", "0", "1", "2"]
    children, root, brlen = db.tree_arrays(tree, labels)
    rng = np.random.default_rng(5)
    L = 40
    msa = rng.integers(0, 5, size=(3, L))
    naive = rng.integers(0, 5, size=L)
    er, pi, rates = [1.0] * 6, np.array([0.17, 0.19, 0.25, 0.39]), orc.gamma_rates_mean(1.0, 4)
    _, _, detail = ao.asr_sample(children, root, brlen, 4, msa, naive, er, pi, rates, 1, 0)
    tips = np.concatenate([naive[None, :], msa], axis=0)
    rows = {lab: i for i, lab in enumerate(labels)}
    for k, r in enumerate(rates):
        ref = orc.per_site_loglik(tree, rows, tips, er, pi, [r])
        np.testing.assert_allclose(detail["loglik_per_rate"][k], ref, rtol=1e-11, atol=1e-12)


def test_committed_oracle_vectors_reproduce():
    """tests/golden/asr_goldens.json (oracle-derived) is what the oracle computes today."""
    import json
    import os
    from tests.golden import make_asr_goldens as mk
    want = json.load(open(os.path.join(os.path.dirname(mk.__file__), "asr_goldens.json")))
    got = json.loads(json.dumps(mk.case()))
    assert got == want

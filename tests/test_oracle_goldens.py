"""Pin the CPU oracle against every literal of the reference's own Catch tests (test/test.cpp),
transcribed into tests/golden/reference_goldens.json by tests/golden/make_goldens.py."""
import os

import numpy as np
import pytest
import yaml

from oracle import linearham_oracle as orc
from tests.helpers import (assert_close_struct, catch_approx, eigen_is_approx, oracle_accessors)

SAMPLE_KEYS = ["naive_seq_samp", "vgerm_state_str_samp", "vgerm_state_ind_samp", "vgerm_left_del_samp",
               "vgerm_right_del_samp", "vgerm_left_insertion_samp", "vd_junction_state_str_samps",
               "vd_junction_state_ind_samps", "vd_junction_insertion_samp", "dgerm_state_str_samp",
               "dgerm_state_ind_samp", "dgerm_left_del_samp", "dgerm_right_del_samp",
               "dj_junction_state_str_samps", "dj_junction_state_ind_samps", "dj_junction_insertion_samp",
               "jgerm_state_str_samp", "jgerm_state_ind_samp", "jgerm_left_del_samp",
               "jgerm_right_del_samp", "jgerm_right_insertion_samp"]


def _load(data_dir, sub, fn):
    with open(os.path.join(data_dir, sub, fn)) as f:
        return yaml.safe_load(f)


@pytest.mark.parametrize("g", ["V", "D", "J"])
def test_germline_parsing(goldens, data_dir, g):
    # test/test.cpp:27-229
    gg = orc.GermlineGene(_load(data_dir, "hmm_params", "IGH%s_ex_star_01.yaml" % g), g)
    want = goldens["Germline"]["vars"]
    for key in ["landing_in", "landing_out", "transition", "emission", "bases"]:
        assert np.array_equal(np.asarray(getattr(gg, key)), np.asarray(want["%s_%s" % (g, key)])), key
    assert gg.gene_prob == want[g + "_gene_prob"]
    assert gg.alphabet == want[g + "_alphabet"]
    assert gg.name == want[g + "_name"]
    assert gg.length == want[g + "_length"]
    if g in "DJ":
        w = goldens["NTInsertion"]["vars"]
        for key in ["nti_landing_in", "nti_landing_out", "nti_transition", "nti_emission"]:
            assert np.array_equal(getattr(gg, key), np.asarray(w["%s_%s" % (g, key)])), key
    if g in "VJ":
        w = goldens["NPadding"]["vars"]
        assert gg.n_transition == w[g + "_n_transition"]
        assert np.array_equal(gg.n_emission, np.asarray(w[g + "_n_emission"]))


def _check_common(h, want):
    got = oracle_accessors(h)
    for k, v in got.items():
        if k not in want:
            continue
        # transition literals are products such as 0.035*0.2*0.1: allow last-bit differences
        tol = 1e-15 if k.endswith("_transition") else 0.0
        assert_close_struct(v, want[k], k, rtol=tol, atol=0.0)


def _check_forward_and_sample(h, want):
    ll = h.log_likelihood()
    assert catch_approx(ll, want["loglikelihood"]), (ll, want["loglikelihood"])
    assert h.vgerm_scaler_count == want["vgerm_scaler_count"]
    assert h.vd_junction_scaler_counts == want["vd_junction_scaler_counts"]
    assert h.dgerm_scaler_count == want["dgerm_scaler_count"]
    assert h.dj_junction_scaler_counts == want["dj_junction_scaler_counts"]
    assert h.jgerm_scaler_count == want["jgerm_scaler_count"]
    h.sample_naive_sequence()
    for k in SAMPLE_KEYS:
        assert h.sample[k] == want[k], (k, h.sample[k], want[k])
    return ll


@pytest.mark.parametrize("case", ["simple_hmm_input", "simple_hmm_input_extra"])
def test_simple_hmm(goldens, data_dir, case):
    # test/test.cpp:245-745
    sec = goldens["SimpleHMM:" + case]
    h = orc.SimpleHMM(os.path.join(data_dir, case + ".yaml"), 0, os.path.join(data_dir, "hmm_params"), 0)
    _check_common(h, sec["vars"])
    assert h.cache_forward == sec["vars"]["cache_forward"]
    ll = _check_forward_and_sample(h, sec["vars"])
    assert abs(ll - sec["vars"]["loglikelihood"]) < 1e-9   # goldens carry 12 significant digits


@pytest.mark.parametrize("case", ["phylo_hmm_input", "phylo_hmm_input_extra"])
def test_phylo_hmm(goldens, data_dir, case):
    # test/test.cpp:750-1368
    sec = goldens["PhyloHMM:" + case]
    meta, want = sec["meta"], sec["vars"]
    h = orc.PhyloHMM(os.path.join(data_dir, case + ".yaml"), 0, os.path.join(data_dir, "hmm_params"), 0)
    h.initialize_phylo_parameters(os.path.join(data_dir, "newton.tree"), meta["er"], meta["pi"],
                                  meta["alpha"], meta["num_rates"])
    h.initialize_phylo_emission()
    _check_common(h, want)
    assert h.xmsa.tolist() == want["xmsa"]
    assert h.xmsa_labels == want["xmsa_labels"]
    assert h.xmsa_seqs == want["xmsa_seqs"]
    assert h.xmsa_naive_ind == want["xmsa_naive_ind"]
    for k in ["vpadding_xmsa_inds", "vgerm_xmsa_inds", "vd_junction_xmsa_inds", "dgerm_xmsa_inds",
              "dj_junction_xmsa_inds", "jgerm_xmsa_inds", "jpadding_xmsa_inds"]:
        assert np.asarray(getattr(h, k)).tolist() == want[k], k
    assert eigen_is_approx(h.xmsa_emission, want["xmsa_emission"], 1e-5)
    # the goldens are printed with 6 significant digits: every entry must agree to that rounding
    rel = np.abs(h.xmsa_emission - np.asarray(want["xmsa_emission"])) / np.asarray(want["xmsa_emission"])
    assert rel.max() < 5e-6, rel.max()
    _check_forward_and_sample(h, want)


def test_phylo_hmm_higher_precision_digits():
    """Digits derived by the survey's independent numpy probe; NOT reference
    goldens, only a tighter self-consistency check of this restatement."""
    d = os.path.join(os.path.dirname(__file__), "golden", "data")
    ll, h = orc.phylo_loglik(os.path.join(d, "phylo_hmm_input.yaml"), os.path.join(d, "hmm_params"),
                             os.path.join(d, "newton.tree"), [1.0] * 6, [0.17, 0.19, 0.25, 0.39], 1.0, 4)
    assert abs(ll - (-75.813635171)) < 1e-8
    np.testing.assert_allclose(h.xmsa_emission[:4], [0.007344744464996, 0.023312167052007,
                                                    0.005637292441960, 0.010786629598697], rtol=1e-12)
    np.testing.assert_allclose(h.sr, [0.136954, 0.476752, 1.0, 2.386294], rtol=2e-6)


def test_phylomd_likelihood(goldens, data_dir):
    # test/test.cpp:1370-1398: pruning alone, cross-checked by the R package phylomd
    sec = goldens["PhyloHMM:phylo_likelihood_hmm_input"]
    meta = sec["meta"]
    ll, _ = orc.phylo_loglik(os.path.join(data_dir, "phylo_likelihood_hmm_input.yaml"),
                             os.path.join(data_dir, "phylo_likelihood_hmm_params"),
                             os.path.join(data_dir, "newton.tree"), meta["er"], meta["pi"], meta["alpha"],
                             meta["num_rates"])
    assert catch_approx(ll, sec["vars"]["loglikelihood"])
    assert abs(ll - (-55.73483)) < 5e-6

"""CPU tests of the C++ host (liblinearham_host.so): parameter parsing, state space, dense transition
matrices and xMSA structures against the reference's Catch-test literals (test/test.cpp) and, on a
synthetic multi-allele family, against the oracle.  No GPU needed: these objects are host-only."""
import os

import numpy as np
import pytest

from linearham_amd import host
from oracle import linearham_oracle as orc
from tests.helpers import assert_close_struct, oracle_accessors

HMM_KEYS_EXACT = ["locus", "flexbounds", "relpos", "alphabet", "msa"]


def _compare_to_golden(dump, want):
    checked = 0
    for k, v in dump.items():
        if k not in want:
            continue
        tol = 1e-15 if k.endswith("_transition") else 0.0
        assert_close_struct(v, want[k], k, rtol=tol)
        checked += 1
    return checked


@pytest.mark.parametrize("g", ["V", "D", "J"])
def test_germline_parsing(goldens, data_dir, g):
    j = host.germline_json(os.path.join(data_dir, "hmm_params", "IGH%s_ex_star_01.yaml" % g), g)
    want = goldens["Germline"]["vars"]
    for key in ["landing_in", "landing_out", "transition", "emission", "bases", "gene_prob", "alphabet", "name",
                "length"]:
        assert_close_struct(j[key], want["%s_%s" % (g, key)], key)
    if g in "DJ":
        for key in ["nti_landing_in", "nti_landing_out", "nti_transition", "nti_emission"]:
            assert_close_struct(j[key], goldens["NTInsertion"]["vars"]["%s_%s" % (g, key)], key)
    if g in "VJ":
        for key in ["n_transition", "n_emission"]:
            assert_close_struct(j[key], goldens["NPadding"]["vars"]["%s_%s" % (g, key)], key)


@pytest.mark.parametrize("case", ["simple_hmm_input", "simple_hmm_input_extra"])
def test_simple_hmm_state_space_and_transitions(goldens, data_dir, case):
    h = host.SimpleHMM(os.path.join(data_dir, case + ".yaml"), 0, os.path.join(data_dir, "hmm_params"), 0)
    n = _compare_to_golden(h.dump(1), goldens["SimpleHMM:" + case]["vars"])
    assert n >= 55


@pytest.mark.parametrize("case", ["phylo_hmm_input", "phylo_hmm_input_extra"])
def test_phylo_hmm_state_space_transitions_xmsa(goldens, data_dir, case):
    h = host.PhyloHMM(os.path.join(data_dir, case + ".yaml"), 0, os.path.join(data_dir, "hmm_params"), 0)
    d = h.dump(1 | 8)
    want = dict(goldens["PhyloHMM:" + case]["vars"])
    want.pop("xmsa_emission")          # needs the GPU; checked in test_host_gpu.py
    want.pop("er"), want.pop("pi")
    want["cache_forward"] = False      # nothing evaluated yet
    d = {k: v for k, v in d.items() if k not in ("xmsa_emission", "er", "pi", "sr", "alpha")}
    n = _compare_to_golden(d, want)
    assert n >= 65


@pytest.mark.parametrize("locus", ["igh", "igk"])
def test_synthetic_family_host_matches_oracle(tmp_path, locus):
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec.small(locus=locus), out)
    yaml_path, pdir = os.path.join(out, "cluster.yaml"), os.path.join(out, "hmm_params")
    o = orc.PhyloHMM(yaml_path, 0, pdir, 0)
    h = host.PhyloHMM(yaml_path, 0, pdir, 0)
    d = h.dump(1 | 8)
    want = oracle_accessors(o)
    for k, v in want.items():
        assert_close_struct(d[k], v, k, rtol=1e-15 if k.endswith("_transition") else 0.0)
    assert d["xmsa"] == o.xmsa.tolist()
    assert d["xmsa_seqs"] == o.xmsa_seqs
    for k in ["vpadding_xmsa_inds", "vgerm_xmsa_inds", "vd_junction_xmsa_inds", "dgerm_xmsa_inds",
              "dj_junction_xmsa_inds", "jgerm_xmsa_inds", "jpadding_xmsa_inds"]:
        assert d[k] == np.asarray(getattr(o, k)).tolist(), k
    # the flattened device inputs (Newick ingest + lh_schedule_tree) need no GPU either
    flat = h.flatten_tsv(os.path.join(out, "trees.tsv"), 7, need_family=False)
    assert flat["n_rows"] == 5 and flat["ops"].shape == (7, 7, 4)
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    np.testing.assert_allclose(flat["alpha"][:5], [r["alpha"] for r in rows])
    assert np.array_equal(flat["ops"][5], flat["ops"][0])
    for i, r in enumerate(rows):
        t = orc.parse_newick(r["tree"])
        total = sum(l for a in t.adj for _, l in a) / 2
        assert abs(flat["brlen"][i].sum() - total) < 1e-12
    if locus == "igh":
        # the table through a FIFO (no size to stat: the reader falls back to a sequential read, as the reference's
        # streaming csv reader would): the same arrays
        import threading
        fifo = str(tmp_path / "trees.fifo")
        os.mkfifo(fifo)
        data = open(os.path.join(out, "trees.tsv"), "rb").read()

        def feed():
            with open(fifo, "wb") as f:
                f.write(data)
        th = threading.Thread(target=feed)
        th.start()
        piped = h.flatten_tsv(fifo, 7, need_family=False)
        th.join()
        for k in ("ops", "brlen", "er", "pi", "alpha"):
            assert np.array_equal(piped[k], flat[k]), k


def test_host_errors(data_dir, tmp_path):
    with pytest.raises(RuntimeError, match="does not exist"):
        host.PhyloHMM(os.path.join(data_dir, "phylo_hmm_input.yaml"), 0, str(tmp_path / "nope"), 0)
    bad = tmp_path / "bad.yaml"
    bad.write_text('{"germline-info": {"locus": "igh"}, "events": [{"unique_ids": ["a"]}]}')
    with pytest.raises(RuntimeError, match="flexbounds"):
        host.PhyloHMM(str(bad), 0, os.path.join(data_dir, "hmm_params"), 0)
    h = host.PhyloHMM(os.path.join(data_dir, "phylo_hmm_input.yaml"), 0, os.path.join(data_dir, "hmm_params"), 0)
    for tree in ["((0:0.2,1:0.4):0.6,naive:0.3);",               # a tip is missing
                 "((0:0.2,1:0.4):0.6,naive:0.3,3:0.5);",          # unknown label
                 "((0:0.2,1:0.4,2:0.1):0.6,naive:0.3);",          # multifurcation below the top
                 "((0:0.2,1:0.4):0.6,naive:0.3,2:0.5"]:           # truncated
        with pytest.raises(RuntimeError):
            h.initialize_phylo_parameters(tree, [1.0] * 6, [0.25] * 4, 1.0, 4, is_path=False)
    # --asr input checks (all before any device work)
    with pytest.raises(RuntimeError, match="Can't open linearham output file"):
        h.run_asr(str(tmp_path / "missing.tsv"), str(tmp_path / "o.trees"), 0)
    cols = "er[1]\ter[2]\ter[3]\ter[4]\ter[5]\ter[6]\tpi[1]\tpi[2]\tpi[3]\tpi[4]\ttree"
    tsv = tmp_path / "bad.tsv"
    tsv.write_text(cols + "\tsr[1]\n")
    with pytest.raises(RuntimeError, match='Missing column "NaiveSequence"'):
        h.run_asr(str(tsv), str(tmp_path / "o.trees"), 0)
    tsv.write_text(cols + "\tNaiveSequence\n")
    with pytest.raises(RuntimeError, match="Missing column .sr.1.."):
        h.run_asr(str(tsv), str(tmp_path / "o.trees"), 0)
    tsv.write_text(cols + "\tsr[1]\tNaiveSequence\n"
                   "1\t1\t1\t1\t1\t1\t.25\t.25\t.25\t.25\t((0:0.2,1:0.4):0.6,naive:0.3,2:0.5);\t1\tACGT\n")
    with pytest.raises(RuntimeError, match="NaiveSequence length differs"):
        h.run_asr(str(tsv), str(tmp_path / "o.trees"), 0)


def test_yaml_reader_dialects(tmp_path, data_dir):
    """Wrapped flow maps, comments and `key:` followed by a same-indent sequence (partis/PyYAML dumps)."""
    src = open(os.path.join(data_dir, "hmm_params", "IGHD_ex_star_01.yaml")).read()
    wrapped = src.replace("insert_left_G: 0.05, insert_left_T: 0.025}",
                          "insert_left_G: 0.05,\n    insert_left_T: 0.025}  # wrapped by the dumper")
    p = tmp_path / "IGHD_ex_star_01.yaml"
    p.write_text("# leading comment\n" + wrapped)
    a = host.germline_json(str(p), "D")
    b = host.germline_json(os.path.join(data_dir, "hmm_params", "IGHD_ex_star_01.yaml"), "D")
    assert a == b


def test_sparse_draw_equals_discrete_distribution():
    """HMM sampling draws with a sparse restatement of libstdc++'s std::discrete_distribution (only the
    non-zero weights are visited): same index, same generator state, on 200 000 random and corner-case
    weight vectors (C++ self-test in liblinearham_host.so)."""
    import ctypes as C
    from linearham_amd import host
    lib = host.load_host()
    lib.lhh_selftest_sparse_draw.argtypes = [C.c_int, C.c_int]
    lib.lhh_selftest_sparse_draw.restype = C.c_int
    assert lib.lhh_selftest_sparse_draw(7, 200000) == 0


def test_newick_export_is_libpll_order(data_dir, tmp_path):
    """The output table's tree column (pll_utree_export_newick, src/PhyloHMM.cpp:299-300): the input's own
    nesting and order, "%f" lengths, `[&index=N]` comments gone, missing / zero lengths replaced by 1e-6.
    Known answers written out by hand from libpll's export format, then host == oracle on synthetic
    RevBayes rows."""
    from linearham_amd import host
    from oracle import linearham_oracle as orc
    from tools import synth_family as sf
    newton = open(os.path.join(data_dir, "newton.tree")).read()
    want = "((0:0.200000,1:0.400000):0.600000,naive:0.300000,2:0.500000);"
    assert orc.export_newick(newton) == want
    assert host.newick_roundtrip(newton, ["naive", "0", "1", "2"]) == want
    odd = "(naive[&index=2]:0.008054984868,s0[&index=3]:0,(s3[&index=5],s4:1e-7)lab[&index=4]:1.5e-3)[&index=1]:0.0;"
    want = "(naive:0.008055,s0:0.000001,(s3:0.000001,s4:0.000000)lab:0.001500);"
    assert orc.export_newick(odd) == want
    assert host.newick_roundtrip(odd, ["naive", "s0", "s3", "s4"]) == want
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec.small(n_samples=6), out)
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    labels = orc.parse_newick(rows[0]["tree"]).labels
    labels = ["naive"] + sorted(l for l in labels if l != "naive")
    for r in rows:
        assert host.newick_roundtrip(r["tree"], labels) == orc.export_newick(r["tree"])
    # a rooted (bifurcating) top level has no libpll order (its unrooted parser rejects it): same tree, unrooted
    rooted = "((naive:0.1,a:0.2):0.05,(b:0.3,c:0.4):0.05);"
    t1 = orc.parse_newick(host.newick_roundtrip(rooted, ["naive", "a", "b", "c"]))
    t2 = orc.parse_newick(rooted)
    assert abs(sum(l for a in t1.adj for _, l in a) - sum(l for a in t2.adj for _, l in a)) < 1e-5


def test_shm_indel_sequences_come_from_indel_reversed_seqs(tmp_path):
    """src/HMM.cpp:74-79: a sequence flagged has_shm_indels is read from indel_reversed_seqs, not input_seqs (which then
    holds the read as sequenced -- here one base short, not alignable as it stands).  Host and oracle build the same MSA
    as for the unflagged family, and the xMSA structures that follow from it."""
    from tools import synth_family as sf
    plain, flagged = str(tmp_path / "plain"), str(tmp_path / "flagged")
    sf.generate(sf.Spec.small(seed=44), plain)
    sf.generate(sf.Spec.small(seed=44, shm_indels=3), flagged)
    import yaml
    ev = yaml.safe_load(open(os.path.join(flagged, "cluster.yaml")))["events"][0]
    assert ev["has_shm_indels"][:4] == [True, True, True, False]
    assert len(ev["input_seqs"][0]) == len(ev["naive_seq"]) - 1 and len(ev["indel_reversed_seqs"][0]) == len(ev["naive_seq"])
    ref = host.PhyloHMM(os.path.join(plain, "cluster.yaml"), 0, os.path.join(plain, "hmm_params"), 0).dump(1 | 8)
    h = host.PhyloHMM(os.path.join(flagged, "cluster.yaml"), 0, os.path.join(flagged, "hmm_params"), 0).dump(1 | 8)
    o = orc.PhyloHMM(os.path.join(flagged, "cluster.yaml"), 0, os.path.join(flagged, "hmm_params"), 0)
    assert h["msa"] == ref["msa"] == o.msa.tolist()
    assert h["xmsa"] == ref["xmsa"] == o.xmsa.tolist()


@pytest.mark.parametrize("kw", [dict(ragged=6, ambiguous=0.02), dict(locus="igk", ragged=5, ambiguous=0.02)])
def test_n_inside_alignment_columns_host_matches_oracle(tmp_path, kw):
    """Ragged reads / ambiguous bases (N inside alignment columns; src/HMM.cpp:69-83, src/utils.cpp:155-164): the host's
    MSA, xMSA (naive base x column pairs, N rows included) and index arrays equal the oracle's."""
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec.small(n_leaves=20, seed=42, **kw), out)
    yaml_path, pdir = os.path.join(out, "cluster.yaml"), os.path.join(out, "hmm_params")
    o = orc.PhyloHMM(yaml_path, 0, pdir, 0)
    n_n = (o.msa == 4).sum(axis=0)
    assert ((n_n > 0) & (n_n < o.msa.shape[0])).sum() >= 10          # columns that mix N with bases
    d = host.PhyloHMM(yaml_path, 0, pdir, 0).dump(1 | 8)
    assert d["msa"] == o.msa.tolist() and d["xmsa"] == o.xmsa.tolist() and d["xmsa_seqs"] == o.xmsa_seqs
    for k in ["vpadding_xmsa_inds", "vgerm_xmsa_inds", "vd_junction_xmsa_inds", "jgerm_xmsa_inds", "jpadding_xmsa_inds"]:
        assert d[k] == np.asarray(getattr(o, k)).tolist(), k

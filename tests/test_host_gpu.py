"""GPU tests of the product path end to end: C++ host classes (linearham's SimpleHMM / PhyloHMM
surface) -> C ABI -> HIP kernels, against the reference's goldens and the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from linearham_amd import host
from oracle import linearham_oracle as orc
from tests.helpers import catch_approx, eigen_is_approx
from tests.test_oracle_goldens import SAMPLE_KEYS

pytestmark = pytest.mark.gpu
ER, PI = [1.0] * 6, [0.17, 0.19, 0.25, 0.39]
FWD_KEYS = ["vgerm_forward", "vd_junction_forward", "dgerm_forward", "dj_junction_forward", "jgerm_forward"]
CNT_KEYS = ["vgerm_scaler_count", "vd_junction_scaler_counts", "dgerm_scaler_count", "dj_junction_scaler_counts",
            "jgerm_scaler_count"]


def _check_against_oracle(h, o, rtol=1e-9):
    ll = h.log_likelihood()
    ref = o.log_likelihood()
    assert abs(ll - ref) <= 1e-12 * abs(ref), (ll, ref)
    d = h.dump(2)
    for k in FWD_KEYS:
        np.testing.assert_allclose(np.asarray(d[k], dtype=float), getattr(o, k), rtol=rtol, atol=0, err_msg=k)
    for k in CNT_KEYS:
        assert d[k] == getattr(o, k), k
    return ll


@pytest.mark.parametrize("case", ["simple_hmm_input", "simple_hmm_input_extra"])
def test_simple_hmm(goldens, data_dir, case):
    # BASELINE.json configs[0]; test/test.cpp:457,718 and the seed-0 sampled paths :377-399,640-660
    want = goldens["SimpleHMM:" + case]["vars"]
    yaml_path, pdir = os.path.join(data_dir, case + ".yaml"), os.path.join(data_dir, "hmm_params")
    h = host.SimpleHMM(yaml_path, 0, pdir, 0)
    o = orc.SimpleHMM(yaml_path, 0, pdir, 0)
    assert h.dump(1)["cache_forward"] is True
    ll = _check_against_oracle(h, o)
    assert catch_approx(ll, want["loglikelihood"]) and abs(ll - want["loglikelihood"]) < 1e-9
    for k in CNT_KEYS:
        assert h.dump(2)[k] == want[k]
    assert h.sample_naive_sequence() == want["naive_seq_samp"]
    s = h.dump(4)
    for k in SAMPLE_KEYS:
        assert s[k] == want[k], k


@pytest.mark.parametrize("case,params,R", [("phylo_hmm_input", "hmm_params", 4),
                                          ("phylo_hmm_input_extra", "hmm_params", 4),
                                          ("phylo_likelihood_hmm_input", "phylo_likelihood_hmm_params", 1)])
def test_phylo_hmm(goldens, data_dir, case, params, R):
    # BASELINE.json configs[1]; test/test.cpp:750-1398
    want = goldens["PhyloHMM:" + case]["vars"]
    yaml_path, pdir = os.path.join(data_dir, case + ".yaml"), os.path.join(data_dir, params)
    tree = os.path.join(data_dir, "newton.tree")
    h = host.PhyloHMM(yaml_path, 0, pdir, 0)
    h.initialize_phylo_parameters(tree, ER, PI, 1.0, R)
    h.initialize_phylo_emission()
    o = orc.PhyloHMM(yaml_path, 0, pdir, 0)
    o.initialize_phylo_parameters(tree, ER, PI, 1.0, R)
    o.initialize_phylo_emission()
    assert h.dump(1)["cache_forward"] is True
    ll = _check_against_oracle(h, o)
    assert catch_approx(ll, want["loglikelihood"])
    d = h.dump(8)
    np.testing.assert_allclose(d["xmsa_emission"], o.xmsa_emission, rtol=1e-10)
    np.testing.assert_allclose(d["sr"], o.sr, rtol=1e-10)
    if "xmsa_emission" in want:
        assert eigen_is_approx(d["xmsa_emission"], want["xmsa_emission"], 1e-5)
        assert h.sample_naive_sequence() == want["naive_seq_samp"]
        s = h.dump(4)
        for k in SAMPLE_KEYS:
            assert s[k] == want[k], k


@pytest.mark.parametrize("locus,n_rows,kw", [
    ("igh", 9, {}), ("igk", 9, {}), ("igh", 150, {}), ("igh", 1, {}), ("igk", 2, {}),
    # ragged reads and ambiguous bases: N inside alignment columns (src/HMM.cpp:69-83, src/PhyloHMM.cpp:368-370)
    ("igh", 12, dict(n_leaves=20, seed=42, ragged=6, ambiguous=0.02)),
    ("igk", 12, dict(n_leaves=12, seed=43, ragged=5, ambiguous=0.02)),
    # more than 64 alleles per segment: the sampled STATES of K4's multi-chunk loops against the oracle's draws, not
    # only against the host sampler (src/HMM.cpp:1222-1353)
    ("igh", 50, dict(n_v=300, n_d=70, n_j=5, seed=21)),
    ("igk", 30, dict(n_v=150, n_j=70, seed=22)),
    # sequences flagged has_shm_indels are read from indel_reversed_seqs (src/HMM.cpp:74-79)
    ("igh", 6, dict(seed=44, shm_indels=3)),
], ids=["igh9", "igk9", "igh150", "igh1", "igk2", "igh_mixed_n", "igk_mixed_n", "many_alleles", "many_alleles_igk", "shm_indels"])
def test_run_pipeline_matches_oracle(tmp_path, locus, n_rows, kw):
    """PhyloHMM::RunPipeline (src/PhyloHMM.cpp:393-446) on a synthetic RevBayes table: batched GPU
    evaluation + sampling (device sampler K4; first and last row re-drawn by the host sampler) must reproduce the
    row-by-row oracle, including the RNG stream -- also when the rows are formatted by several worker threads (150
    rows: every core of the box takes a share)."""
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec.small(n_samples=n_rows, locus=locus, **kw), out)
    yaml_path, pdir, tsv = os.path.join(out, "cluster.yaml"), os.path.join(out, "hmm_params"), os.path.join(out, "trees.tsv")
    h = host.PhyloHMM(yaml_path, 0, pdir, 3)
    res = os.path.join(out, "lh.tsv")
    h.run_pipeline(tsv, res, 4)
    lines = [l.rstrip("\n").split("\t") for l in open(res)]
    header, body = lines[0], lines[1:]
    col = {name: i for i, name in enumerate(header)}
    assert header[:4] == ["Iteration", "RBLogLikelihood", "Prior", "alpha"] and header[-1] == "JFwkInsertion"
    o = orc.PhyloHMM(yaml_path, 0, pdir, 3)
    rows = sf.read_trees_tsv(tsv)
    assert len(body) == len(rows)
    for r, got in zip(rows, body):
        o.initialize_phylo_parameters(r["tree"], r["er"], r["pi"], r["alpha"], 4, is_path=False)
        o.initialize_phylo_emission()
        ll = o.log_likelihood()
        seq = o.sample_naive_sequence()
        assert int(got[col["Iteration"]]) == r["iteration"]
        assert abs(float(got[col["LHLogLikelihood"]]) - ll) <= 5e-6 * abs(ll)        # 6 significant digits
        assert abs(float(got[col["LogWeight"]]) - (ll - r["likelihood"])) <= 5e-6 * abs(ll - r["likelihood"]) + 1e-9
        assert got[col["NaiveSequence"]] == seq
        assert got[col["VGene"]] == o.sample["vgerm_state_str_samp"]
        assert got[col["JGene"]] == o.sample["jgerm_state_str_samp"]
        dels = [("vgerm_left_del_samp", "V5pDel"), ("vgerm_right_del_samp", "V3pDel"),
                ("jgerm_left_del_samp", "J5pDel"), ("jgerm_right_del_samp", "J3pDel")]
        if locus == "igh":
            assert got[col["DGene"]] == o.sample["dgerm_state_str_samp"]
            assert got[col["VDInsertion"]] == o.sample["vd_junction_insertion_samp"]
            assert got[col["DJInsertion"]] == o.sample["dj_junction_insertion_samp"]
            dels += [("dgerm_left_del_samp", "D5pDel"), ("dgerm_right_del_samp", "D3pDel")]
        else:
            assert "DGene" not in col
            assert got[col["VJInsertion"]] == o.sample["vd_junction_insertion_samp"]
        for k, c in dels:
            assert int(got[col[c]]) == o.sample[k], c
        np.testing.assert_allclose([float(got[col["sr[%d]" % i]]) for i in range(1, 5)], o.sr, rtol=5e-6)
        # the tree column is the input tree as libpll would re-export it
        assert got[col["tree"]] == orc.export_newick(r["tree"])
        t1, t2 = orc.parse_newick(got[col["tree"]]), orc.parse_newick(r["tree"])
        assert abs(sum(l for a in t1.adj for _, l in a) - sum(l for a in t2.adj for _, l in a)) < 1e-4


def _annotated(children, root, brlen, labels, naive_seq, msa, anc, alphabet="ACGTN"):
    """The line PhyloHMM::RunAsr writes (scripts/run_bootstrap_asr_ess.R:86-101), rebuilt from oracle states."""
    T = len(labels)

    def comment(v):
        if v == 0:
            s = naive_seq
        elif v < T:
            s = "".join(alphabet[b] for b in msa[v - 1])
        else:
            s = "".join(alphabet[b] for b in anc[v - T])
        return '[&ancestral="%s"]' % s

    def go(v):
        if v < T:
            return labels[v] + comment(v) + ":%.10g" % brlen[v]
        a, b = children[2 * (v - T)], children[2 * (v - T) + 1]
        return "(" + go(a) + "," + go(b) + ")" + comment(v) + ":%.10g" % (0.0 if v == root else brlen[v])

    return "(" + labels[0] + comment(0) + ":%.10g" % brlen[0] + "," + go(root) + ")" + comment(root) + ";"


@pytest.mark.parametrize("locus", ["igh", "igk"])
def test_run_asr_matches_oracle(tmp_path, locus):
    """--pipeline then --asr (scripts/run_bootstrap_asr_ess.R:48-104 on every row of the pipeline's table): the
    annotated trees must be the oracle's draws (same Philox stream; rates = the 6-digit sr[] columns, naive =
    the NaiveSequence column, as the R script reads them), written in the host's own node numbering."""
    from oracle import asr_oracle as ao
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec.small(n_samples=5, locus=locus, seed=31), out)
    yaml_path, pdir, tsv = os.path.join(out, "cluster.yaml"), os.path.join(out, "hmm_params"), os.path.join(out, "trees.tsv")
    h = host.PhyloHMM(yaml_path, 0, pdir, 3)
    res, asr = os.path.join(out, "lh.tsv"), os.path.join(out, "asr.trees")
    h.run_pipeline(tsv, res, 4)
    h.run_asr(res, asr, 77)
    got = [l.rstrip("\n") for l in open(asr)]
    lines = [l.rstrip("\n").split("\t") for l in open(res)]
    col = {name: i for i, name in enumerate(lines[0])}
    o = orc.PhyloHMM(yaml_path, 0, pdir, 3)
    labels = list(o.xmsa_labels)
    assert len(got) == len(lines) - 1
    for i, f in enumerate(lines[1:]):
        children, root, brlen = host.newick_arrays(f[col["tree"]], labels)
        er = [float(f[col["er[%d]" % k]]) for k in range(1, 7)]
        pi = np.array([float(f[col["pi[%d]" % k]]) for k in range(1, 5)])
        sr = [float(f[col["sr[%d]" % k]]) for k in range(1, 5)]
        naive_seq = f[col["NaiveSequence"]]
        naive = np.array(["ACGTN".index(c) for c in naive_seq])
        _, anc, _ = ao.asr_sample(children, root, brlen, len(labels), o.msa, naive, er, pi, sr, 77, i)
        want = _annotated(children, root, brlen, labels, naive_seq, o.msa, anc)
        assert got[i] == want, i


def test_cli_compute_logl(data_dir):
    """`linearham --compute-logl` prints the log-likelihood with 6 significant digits
    (src/linearham.cpp:341-348)."""
    exe = _exe()
    cmd = [exe, "--compute-logl", "--yaml-path", os.path.join(data_dir, "phylo_hmm_input.yaml"), "--cluster-ind", "0",
           "--hmm-param-dir", os.path.join(data_dir, "hmm_params"), "--newick-path",
           os.path.join(data_dir, "newton.tree"), "--num-rates", "4"]
    for x in ER:
        cmd += ["--er", str(x)]
    for x in PI:
        cmd += ["--pi", str(x)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "-75.8136"
    bad = subprocess.run([exe, "--nonsense", "--yaml-path", "x"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "ERROR:" in bad.stderr


def _exe():
    exe = os.path.join(os.path.dirname(host.host_library_path()), "linearham")
    if not os.path.exists(exe):   # a built artefact (not in history): g++ is enough to make it
        from linearham_amd import build as lb
        lb.build_host(verbose=False)
    return exe


def test_cli_sample(goldens, data_dir):
    """`linearham --sample` (src/linearham.cpp:353-356): N naive sequences, one per line; the first with seed 0 is
    the reference's golden draw (test/test.cpp:888-910), the whole list what the library's own calls give."""
    want = goldens["PhyloHMM:phylo_hmm_input"]["vars"]
    yaml_path, pdir = os.path.join(data_dir, "phylo_hmm_input.yaml"), os.path.join(data_dir, "hmm_params")
    tree = os.path.join(data_dir, "newton.tree")
    cmd = [_exe(), "--sample", "--yaml-path", yaml_path, "--cluster-ind", "0", "--hmm-param-dir", pdir, "--newick-path",
           tree, "--num-rates", "4", "--seed", "0", "--N", "5"]
    for x in ER:
        cmd += ["--er", str(x)]
    for x in PI:
        cmd += ["--pi", str(x)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = out.stdout.split()
    assert len(got) == 5 and got[0] == want["naive_seq_samp"]
    h = host.PhyloHMM(yaml_path, 0, pdir, 0)
    h.initialize_phylo_parameters(tree, ER, PI, 1.0, 4)
    h.initialize_phylo_emission()
    assert got == [h.sample_naive_sequence() for _ in range(5)]


def test_cli_pipeline(tmp_path):
    """`linearham --pipeline` (src/linearham.cpp:390-422) writes, byte for byte, the table the library call writes
    (which test_run_pipeline_matches_oracle pins to the oracle), and the extended-range switch leaves a table the
    reference evaluates without overflow unchanged."""
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec.small(n_samples=40, seed=17), out)
    yaml_path, pdir, tsv = os.path.join(out, "cluster.yaml"), os.path.join(out, "hmm_params"), os.path.join(out, "trees.tsv")
    lib_out, cli_out, ext_out = (os.path.join(out, n) for n in ("lib.tsv", "cli.tsv", "ext.tsv"))
    host.PhyloHMM(yaml_path, 0, pdir, 5).run_pipeline(tsv, lib_out, 4)
    common = ["--yaml-path", yaml_path, "--cluster-ind", "0", "--hmm-param-dir", pdir, "--input-path", tsv,
              "--num-rates", "4", "--seed", "5"]
    r = subprocess.run([_exe(), "--pipeline"] + common + ["--output-path", cli_out], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(cli_out).read() == open(lib_out).read()
    assert len(open(cli_out).read().splitlines()) == 41
    r = subprocess.run([_exe(), "--pipeline"] + common + ["--output-path", ext_out, "--extended-range", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(ext_out).read() == open(lib_out).read()
    bad = subprocess.run([_exe(), "--pipeline"] + common[:-4] + ["--input-path", tsv + ".missing", "--output-path",
                                                                  cli_out], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "ERROR:" in bad.stderr


SAMPLER_FAMILIES = {
    "igh3": dict(kind="small", n_rows=5000, kw=dict(n_samples=64, seed=3, locus="igh")),
    "igk4": dict(kind="small", n_rows=5000, kw=dict(n_samples=64, seed=4, locus="igk")),
    "igh11": dict(kind="small", n_rows=5000, kw=dict(n_samples=64, seed=11, locus="igh")),
    # more than 64 alleles per segment: K4's multi-chunk loops over the left genes (lh_sample.hip: 64 genes at a time)
    "many_alleles": dict(kind="small", n_rows=1500, kw=dict(n_samples=32, seed=21, n_v=300, n_d=70, n_j=5)),
    "many_alleles_igk": dict(kind="small", n_rows=1500, kw=dict(n_samples=32, seed=22, locus="igk", n_v=150, n_j=70)),
    # N inside alignment columns (ragged reads, ambiguous bases): K1's N-aware kernels feed the forward arrays K4 draws from
    "mixed_n": dict(kind="small", n_rows=3000, kw=dict(n_samples=32, seed=42, n_leaves=20, ragged=6, ambiguous=0.02)),
    # BASELINE.json configs[2] at full size: 100 leaves x 400 sites, 200 V (four chunks) / 30 D / 12 J
    "config2": dict(kind="full", n_rows=256, kw=dict(n_samples=256)),
}


@pytest.mark.parametrize("name", sorted(SAMPLER_FAMILIES))
def test_device_sampler_matches_host_sampler(tmp_path, name):
    """The states K4 (lh_eval_sample_batch) draws on the device are the ones HMM::SampleRow draws on the host from
    the same forward arrays and the same std::mt19937 stream (src/HMM.cpp:358-431): `linearham --pipeline` writes the
    same bytes with and without LH_HOST_SAMPLING, over several launch batches (5000 rows > RunPipeline's 2048) --
    small families, families with 150-300 alleles per segment, and configs[2] at its full size."""
    from tools import synth_family as sf
    cfg = SAMPLER_FAMILIES[name]
    seed = cfg["kw"].get("seed", 1)
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec.small(**cfg["kw"]) if cfg["kind"] == "small" else sf.Spec(**cfg["kw"]), out)
    yaml_path, pdir, tsv = os.path.join(out, "cluster.yaml"), os.path.join(out, "hmm_params"), os.path.join(out, "trees.tsv")
    lines = open(tsv).read().splitlines()
    big = os.path.join(out, "big.tsv")
    n_rows = cfg["n_rows"]
    with open(big, "w") as f:
        f.write(lines[0] + "\n")
        for i in range(n_rows):
            f.write(lines[1 + i % (len(lines) - 1)] + "\n")
    common = ["--yaml-path", yaml_path, "--cluster-ind", "0", "--hmm-param-dir", pdir, "--input-path", big,
              "--num-rates", "4", "--seed", str(seed)]
    outs = {}
    for mode in ("device", "host"):
        env = dict(os.environ)
        if mode == "host":
            env["LH_HOST_SAMPLING"] = "1"
        else:
            env.pop("LH_HOST_SAMPLING", None)
        env["LH_PIPELINE_TIMING"] = "1"
        o = os.path.join(out, mode + ".tsv")
        r = subprocess.run([_exe(), "--pipeline"] + common + ["--output-path", o], capture_output=True, text=True,
                           timeout=600, env=env)
        assert r.returncode == 0, r.stderr
        outs[mode] = open(o).read()
        assert ("device sampler" in r.stderr) == (mode == "device"), r.stderr
    assert outs["device"] == outs["host"]
    rows = [ln.split("\t") for ln in outs["device"].splitlines()]
    c = rows[0].index("NaiveSequence")
    assert len(rows) == n_rows + 1 and len({r[c] for r in rows[1:]}) > 1   # the draws do vary over the rows


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_pipeline_with_several_handles_writes_the_same_table(tmp_path, devices):
    """`linearham --pipeline --devices a,b,...`: one family handle and one host thread per listed device, table row i
    evaluated and sampled on device i mod N (the loop that shards: src/PhyloHMM.cpp:414-442), output in
    file order.  On the one-GPU box the list names device 0 two and three times -- two / three handles driven from
    their own threads: the table must be byte-identical to the single-handle one (5000 rows, several launch batches,
    shards of unequal size)."""
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec.small(n_samples=64, seed=9), out)
    yaml_path, pdir, tsv = os.path.join(out, "cluster.yaml"), os.path.join(out, "hmm_params"), os.path.join(out, "trees.tsv")
    lines = open(tsv).read().splitlines()
    big = os.path.join(out, "big.tsv")
    with open(big, "w") as f:
        f.write(lines[0] + "\n")
        for i in range(5000):
            f.write(lines[1 + i % (len(lines) - 1)] + "\n")
    common = ["--yaml-path", yaml_path, "--cluster-ind", "0", "--hmm-param-dir", pdir, "--input-path", big,
              "--num-rates", "4", "--seed", "9"]
    tables = {}
    for name, extra in (("one", []), ("many", ["--devices", devices])):
        o = os.path.join(out, name + ".tsv")
        r = subprocess.run([_exe(), "--pipeline"] + common + extra + ["--output-path", o], capture_output=True, text=True,
                           timeout=600, env=dict(os.environ, LH_PIPELINE_TIMING="1"))
        assert r.returncode == 0, r.stderr
        assert "device sampler" in r.stderr, r.stderr
        tables[name] = open(o).read()
    assert tables["one"] == tables["many"] and tables["one"].count("\n") == 5001
    bad = subprocess.run([_exe(), "--pipeline"] + common + ["--devices", "0,7", "--output-path", os.path.join(out, "x.tsv")],
                         capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "no such device" in bad.stderr      # a device the box does not have
    # a run that samples on the host does not shard: it says so (instead of building handles nobody uses) and writes the
    # same table on the first listed device
    o = os.path.join(out, "hostsampling.tsv")
    r = subprocess.run([_exe(), "--pipeline"] + common + ["--devices", devices, "--output-path", o], capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, LH_HOST_SAMPLING="1"))
    assert r.returncode == 0, r.stderr
    assert "samples on the host" in r.stderr, r.stderr
    assert open(o).read() == tables["one"]


@pytest.mark.parametrize("locus", ["igh", "igk", "many_alleles", "many_alleles_igk"])
def test_device_sampler_on_crafted_engine_outputs(tmp_path, locus):
    """K4 against HMM::SampleRow draw by draw on engine outputs chosen to hit the corners of libstdc++'s
    discrete_distribution (bits/random.tcc): a uniform of exactly 0 (lower_bound returns element 0 whatever it holds),
    the largest uniform below 1 (the last partial sum is forced to 1), uniforms next to 0 and 1, and random words.
    The host engine is a std::mt19937 whose state is set so that it returns exactly these words."""
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    kw = {"igh": dict(locus="igh"), "igk": dict(locus="igk"), "many_alleles": dict(n_v=300, n_d=70, n_j=5),
          "many_alleles_igk": dict(locus="igk", n_v=150, n_j=70)}[locus]   # the last two: K4's loops over > 64 genes
    sf.generate(sf.Spec.small(n_samples=3, seed=23, **kw), out)
    yaml_path, pdir = os.path.join(out, "cluster.yaml"), os.path.join(out, "hmm_params")
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    h = host.PhyloHMM(yaml_path, 0, pdir, 0)
    rng = np.random.default_rng(5)
    n_words = 600
    patterns = [np.zeros(n_words, np.uint32), np.full(n_words, 0xFFFFFFFF, np.uint32)]
    lo1 = np.zeros(n_words, np.uint32); lo1[0::2] = 1                    # smallest positive uniform
    hi1 = np.full(n_words, 0xFFFFFFFF, np.uint32); hi1[0::2] = 0xFFFFF800  # rounds to just below 1
    mix = np.where(rng.random(n_words) < 0.5, 0, 0xFFFFFFFF).astype(np.uint32)
    patterns += [lo1, hi1, mix] + [rng.integers(0, 2 ** 32, n_words, dtype=np.uint64).astype(np.uint32) for _ in range(20)]
    seen = set()
    for r in rows:
        h.initialize_phylo_parameters(r["tree"], r["er"], r["pi"], r["alpha"], 4, is_path=False)
        for w in patterns:
            dev, ref = h.sample_states_with_words(w)
            np.testing.assert_array_equal(dev, ref)
            seen.add(tuple(dev))
    assert len(seen) > 10   # the patterns do lead to different paths


def test_full_size_family_properties(tmp_path):
    """BASELINE.json configs[2] at full size (100 leaves x 400 sites, 200 V / 30 D / 12 J alleles) through the C++
    host and the C ABI -- properties that need no oracle run: rows repeated in a batch give identical bits wherever
    they sit (launch layout independence); swapping the two children of every inner node (another schedule of the
    same tree) changes the log-likelihood by rounding only; the host-pointer and a second call agree bit for bit;
    a handful of rows agree with the dense C oracle to 1e-12."""
    import ctypes as C
    import linearham_amd
    from linearham_amd import host as hst
    from oracle import oracle_c
    from tests import desc_builder as db
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec(n_samples=48), out)
    hmm = hst.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    n = 3000                                    # 62 copies of each row, interleaved
    flat = hmm.flatten_tsv(os.path.join(out, "trees.tsv"), n)
    lib = linearham_amd.load_library()
    T, depth = flat["n_tips"], flat["max_depth"]

    def run(ops):
        ll = np.zeros(n)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        lib.check(lib.lib.lh_eval_batch(C.c_void_p(flat["family"]), n, T, 16, p(np.ascontiguousarray(ops), C.c_int32),
                                        p(flat["brlen"], C.c_double), p(flat["er"], C.c_double),
                                        p(flat["pi"], C.c_double), p(flat["alpha"], C.c_double), 4,
                                        p(ll, C.c_double), None))
        return ll
    ll = run(flat["ops"])
    assert np.all(np.isfinite(ll))
    rows = flat["n_rows"]
    for r in range(rows):
        assert len(set(ll[r::rows].tolist())) == 1, r          # identical bits for identical rows
    assert np.array_equal(ll, run(flat["ops"]))
    # the same trees scheduled with every inner node's children swapped
    rows_tsv = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    o = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    labels = list(o.xmsa_labels)
    ops2 = flat["ops"].copy()
    trees = []
    for r in range(rows):
        children, root, brlen = hst.newick_arrays(rows_tsv[r]["tree"], labels)
        trees.append((children, root, brlen))
        swapped = children.reshape(-1, 2)[:, ::-1].ravel().copy()
        sched, d = lib.schedule_tree(T, swapped, root)
        assert d <= 16
        ops2[r::rows] = sched
    ll2 = run(ops2)
    np.testing.assert_allclose(ll2, ll, rtol=1e-12)
    # a few rows against the dense reference algorithm (C oracle)
    oracle_c.build()
    fam = oracle_c.COracleFamily(o, 4)
    idx = [0, 7, 23]
    ref = fam.eval([trees[i] for i in idx], [rows_tsv[i]["er"] for i in idx], [rows_tsv[i]["pi"] for i in idx],
                   [rows_tsv[i]["alpha"] for i in idx], n_threads=3)
    np.testing.assert_allclose(ll[idx], ref, rtol=1e-12)


def test_config4_full_size_family(tmp_path):
    """BASELINE.json configs[4] at its stated size: 500 leaves x 600 sites with the FULL germline set (200 V /
    30 D / 12 J alleles), through the C++ host and the C ABI.  Layout properties on a whole batch (identical
    rows give identical bits wherever they sit; a second call repeats bit for bit), every distinct row against
    the dense C oracle to 1e-12 -- and, on the tree samples where the reference's own 2^(256 d) equalisation
    overflows (src/PhyloHMM.cpp:190-192), the same non-finite mask on both sides."""
    import ctypes as C
    import linearham_amd
    from linearham_amd import host as hst
    from oracle import oracle_c
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec(n_leaves=500, n_sites=600, n_samples=64), out)
    hmm = hst.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    sizes = hmm.sizes()
    assert sizes["n_tips"] == 501 and sizes["n_sites"] == 600
    n = 704                                     # 11 copies of each row
    flat = hmm.flatten_tsv(os.path.join(out, "trees.tsv"), n)
    lib = linearham_amd.load_library()
    T, depth, rows = flat["n_tips"], flat["max_depth"], flat["n_rows"]
    assert rows == 64 and depth <= 16

    def run():
        ll = np.zeros(n)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        lib.check(lib.lib.lh_eval_batch(C.c_void_p(flat["family"]), n, T, depth, p(flat["ops"], C.c_int32),
                                        p(flat["brlen"], C.c_double), p(flat["er"], C.c_double),
                                        p(flat["pi"], C.c_double), p(flat["alpha"], C.c_double), 4,
                                        p(ll, C.c_double), None))
        return ll
    ll = run()
    for r in range(rows):
        assert len(set(np.nan_to_num(ll[r::rows], nan=1.0, posinf=2.0, neginf=3.0).tolist())) == 1, r
    assert np.array_equal(ll, run(), equal_nan=True)
    # every distinct row against the dense reference algorithm (C oracle)
    rows_tsv = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    o = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    labels = list(o.xmsa_labels)
    trees = [hst.newick_arrays(rows_tsv[r]["tree"], labels) for r in range(rows)]
    oracle_c.build()
    fam = oracle_c.COracleFamily(o, 4)
    ref = fam.eval(trees, [r["er"] for r in rows_tsv], [r["pi"] for r in rows_tsv], [r["alpha"] for r in rows_tsv],
                   n_threads=min(16, len(os.sched_getaffinity(0))))
    got = ll[:rows]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin)               # the reference's overflow rows, and only those
    assert fin.sum() >= rows - 8                                # ... are a small minority
    np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-12)


def test_extended_range_mode(tmp_path):
    """Opt-in extended-range arithmetic (lh_family_set_extended_range; not reference behaviour): on the full
    configs[4] family it must (a) equal the default mode's log-likelihood to 1e-10 wherever that is finite,
    (b) be finite on the tree samples where the reference's 2^(256 d) equalisation overflows, (c) agree there
    with the same mode restated on the dense algorithm (oracle_kernels.c, ext), and (d) leave a sampling
    distribution: forward rows differ from the default's only by a positive factor per row."""
    import ctypes as C
    import linearham_amd
    from linearham_amd import host as hst
    from linearham_amd.capi import _EvalOutputs
    from oracle import oracle_c
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec(n_leaves=500, n_sites=600, n_samples=64), out)
    hmm = hst.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    flat = hmm.flatten_tsv(os.path.join(out, "trees.tsv"), 64)
    lib = linearham_amd.load_library()
    fam = C.c_void_p(flat["family"])
    T, depth, n = flat["n_tips"], flat["max_depth"], 64
    fs = lib.lib.lh_forward_size(fam)

    def run(want_fwd=False):
        ll = np.zeros(n)
        fwd = np.zeros((n, fs)) if want_fwd else None
        outs = _EvalOutputs()
        if want_fwd:
            outs.forward = fwd.ctypes.data_as(C.POINTER(C.c_double))
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        lib.check(lib.lib.lh_eval_batch(fam, n, T, depth, p(flat["ops"], C.c_int32), p(flat["brlen"], C.c_double),
                                        p(flat["er"], C.c_double), p(flat["pi"], C.c_double),
                                        p(flat["alpha"], C.c_double), 4, p(ll, C.c_double), C.byref(outs)))
        return ll, fwd
    ll_def, fwd_def = run(True)
    hmm.set_extended_range(True)
    ll_ext, fwd_ext = run(True)
    hmm.set_extended_range(False)
    ll_back, _ = run()
    assert np.array_equal(ll_back, ll_def, equal_nan=True)          # the switch is clean
    fin = np.isfinite(ll_def)
    assert 0 < (~fin).sum() <= 8                                    # the family does contain overflow rows
    assert np.all(np.isfinite(ll_ext))
    np.testing.assert_allclose(ll_ext[fin], ll_def[fin], rtol=1e-10)
    # the overflow rows against the dense restatement of the mode
    rows_tsv = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    o = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    labels = list(o.xmsa_labels)
    bad = [int(i) for i in np.where(~fin)[0]] + [0, 1]
    trees = [hst.newick_arrays(rows_tsv[r]["tree"], labels) for r in bad]
    oracle_c.build()
    ofam = oracle_c.COracleFamily(o, 4)
    ref = ofam.eval(trees, [rows_tsv[r]["er"] for r in bad], [rows_tsv[r]["pi"] for r in bad],
                    [rows_tsv[r]["alpha"] for r in bad], n_threads=4, extended=True)
    np.testing.assert_allclose(ll_ext[bad], ref, rtol=1e-12)
    # forward rows of a finite sample: same direction in both modes (V germline vector: the first nV entries)
    nV = 200
    a, b = fwd_def[0, :nV], fwd_ext[0, :nV]
    keep = a > a.max() * 1e-200
    ratio = b[keep] / a[keep]
    np.testing.assert_allclose(ratio, ratio[0], rtol=1e-12)

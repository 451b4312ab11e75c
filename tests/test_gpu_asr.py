"""GPU parity tests of the ancestral-sequence sampling kernel (K3, lh_asr_batch) against
oracle/asr_oracle.py, draw by draw (same Philox stream), through the C ABI.

Reference step: scripts/run_bootstrap_asr_ess.R:48-104.  Bit-exact bar: the sampled states and rate
categories are integers; they must be identical except where a uniform lands within rounding distance of a
category boundary (the oracle and the kernel normalise their weights differently), which the tests bound at a
handful of sites per million."""
import os

import numpy as np
import pytest

from oracle import asr_oracle as ao
from oracle import linearham_oracle as orc
from tests import desc_builder as db

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import linearham_amd
    lib = linearham_amd.load_library()
    assert lib.device_count() >= 1, "no HIP device visible: the GPU tests need an MI355X"
    return lib


def _run(hip, h, rows, R, seed, first_sample, rng, naive_with_n=True):
    import linearham_amd
    fam = linearham_amd.Family(db.build_family_desc(h), hip)
    T = h.msa.shape[0] + 1
    L = h.msa.shape[1]
    ops, brl, trees, depth = [], [], [], 0
    for s in rows:
        children, root, brlen = db.tree_arrays(orc.parse_newick(s["tree"]), h.xmsa_labels)
        o, d = hip.schedule_tree(T, children, root)
        ops.append(o)
        brl.append(brlen)
        trees.append((children, root, brlen))
        depth = max(depth, d)
    n = len(rows)
    rates = np.stack([orc.gamma_rates_mean(s["alpha"], R) for s in rows])
    naive = rng.integers(0, 5 if naive_with_n else 4, size=(n, L)).astype(np.uint8)
    anc, choice = fam.asr_batch(T, depth, np.stack(ops), np.stack(brl), [s["er"] for s in rows],
                                [s["pi"] for s in rows], rates, naive, seed, first_sample)
    fam.close()
    mism = 0
    for i, s in enumerate(rows):
        children, root, brlen = trees[i]
        c_ref, a_ref, _ = ao.asr_sample(children, root, brlen, T, h.msa, naive[i], s["er"], np.asarray(s["pi"]),
                                        rates[i], seed, first_sample + i)
        same_rate = c_ref == choice[i]
        mism += int((~same_rate).sum())
        # where the category agrees the joint draw must agree node by node
        mism += int((a_ref[:, same_rate] != anc[i][:, same_rate]).any(axis=0).sum())
        assert anc[i].max() <= 3
    return mism, n * L, anc, choice


@pytest.mark.parametrize("preset", ["small", "medium", "igk", "rates1", "rates8", "mixed_n", "mixed_n_deep"])
def test_asr_matches_oracle(hip, tmp_path, preset):
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    spec = {"small": sf.Spec.small(n_samples=5), "medium": sf.Spec.small(n_leaves=40, n_samples=3, seed=11),
            "igk": sf.Spec.small(locus="igk", n_samples=3, seed=5), "rates1": sf.Spec.small(n_samples=2, seed=8),
            "rates8": sf.Spec.small(n_samples=2, n_leaves=17, seed=9),
            # N inside alignment columns (ragged reads, ambiguous bases): K1's N-aware unfused planes, K3's tip handling of
            # state 4 in the MSA on the way up and down; the second on a balanced tree (stack depth >= 5)
            "mixed_n": sf.Spec.small(n_leaves=20, n_samples=3, seed=42, ragged=6, ambiguous=0.02),
            "mixed_n_deep": sf.Spec.small(n_leaves=64, n_samples=2, seed=49, tree_shape="balanced", n_nni=0, ragged=6,
                                          ambiguous=0.01)}[preset]
    R = {"rates1": 1, "rates8": 8}.get(preset, 4)
    sf.generate(spec, out)
    h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    if preset.startswith("mixed_n"):
        n_n = (h.msa == 4).sum(axis=0)
        assert ((n_n > 0) & (n_n < h.msa.shape[0])).sum() > 10
    mism, total, anc, choice = _run(hip, h, rows, R, seed=20261004, first_sample=7, rng=np.random.default_rng(3))
    assert mism <= max(1, total // 200000), (mism, total)
    assert choice.max() < R


def test_asr_at_the_alpha_floor(hip, tmp_path):
    """alpha = 0.05 with eight categories: the lowest rates are 1e-17 .. 1e-11, a 1e-6 branch's off-diagonal P entries 1e-19 and
    smaller.  The kernels form P = I + U expm1(lambda t r) U^-1 clamped at 0; the oracle's exp() form gave such entries either
    sign and drew columns with mutations into category 0 (round 4, found by tests/dev_tools/random_sweep_asr.py) until it asked
    for the same form.  Every draw must agree, and no column that varies may be drawn into the slowest category."""
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec.small(n_leaves=14, n_samples=3, seed=9005, locus="igk", ambiguous=0.05, tree_shape="balanced"), out)
    h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    rows = [dict(r, alpha=0.05) for r in sf.read_trees_tsv(os.path.join(out, "trees.tsv"))]
    mism, total, anc, choice = _run(hip, h, rows, 8, seed=9005, first_sample=3, rng=np.random.default_rng(5))
    assert mism == 0, (mism, total)
    varies = np.array([len(set(h.msa[:, j][h.msa[:, j] < 4])) > 1 for j in range(h.msa.shape[1])])
    assert varies.sum() >= 3 and (choice[:, varies] > 0).all(), (int(varies.sum()), choice[:, varies].min(axis=0).tolist())


def test_asr_toy_family_and_determinism(hip, data_dir):
    """The reference's own toy family (data/phylo_hmm_input.yaml + newton.tree): smallest tree with an inner
    branch; same call twice -> identical output; different seed -> different output; sample numbering is
    by first_sample + i, not by position in the batch."""
    h = orc.PhyloHMM(os.path.join(data_dir, "phylo_hmm_input.yaml"), 0, os.path.join(data_dir, "hmm_params"), 0)
    tree = open(os.path.join(data_dir, "newton.tree")).read()
    rows = [dict(tree=tree, er=[1.0] * 6, pi=[0.17, 0.19, 0.25, 0.39], alpha=1.0)] * 6
    m1, total, anc1, ch1 = _run(hip, h, rows, 4, 5, 0, np.random.default_rng(1))
    m2, _, anc2, ch2 = _run(hip, h, rows, 4, 5, 0, np.random.default_rng(1))
    assert m1 == 0 and m2 == 0
    assert np.array_equal(anc1, anc2) and np.array_equal(ch1, ch2)
    _, _, anc3, _ = _run(hip, h, rows, 4, 6, 0, np.random.default_rng(1))
    assert not np.array_equal(anc1, anc3)
    _, _, anc4, ch4 = _run(hip, h, rows[:3], 4, 5, 3, np.random.default_rng(1))
    # rows 3..5 of the first call used sample numbers 3..5 but other naive sequences: only shapes compare
    assert anc4.shape == (3, 2, h.msa.shape[1])


def test_asr_large_tree(hip, tmp_path):
    """500 leaves x 600 sites (the LDS tables of one (sample, rate) take ~147 KB), stack depth 4."""
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec(n_leaves=500, n_sites=600, n_v=24, n_d=6, n_j=4, n_samples=2, seed=99), out)
    h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    mism, total, _, _ = _run(hip, h, rows, 4, 99, 0, np.random.default_rng(2))
    assert mism <= 1, (mism, total)


def test_asr_error_paths(hip, data_dir):
    import linearham_amd
    h = orc.PhyloHMM(os.path.join(data_dir, "phylo_hmm_input.yaml"), 0, os.path.join(data_dir, "hmm_params"), 0)
    fam = linearham_amd.Family(db.build_family_desc(h), hip)
    T, L = 4, h.msa.shape[1]
    children, root, brlen = db.tree_arrays(orc.parse_newick(open(os.path.join(data_dir, "newton.tree")).read()),
                                           h.xmsa_labels)
    ops, depth = hip.schedule_tree(T, children, root)
    good = dict(ops=ops[None], brlen=brlen[None], er=[[1.0] * 6], pi=[[0.25] * 4], rates=[[1.0]],
                naive=np.zeros((1, L), dtype=np.uint8))
    fam.asr_batch(T, depth, good["ops"], good["brlen"], good["er"], good["pi"], good["rates"], good["naive"], 1)
    bad = np.full((1, L), 7, dtype=np.uint8)
    with pytest.raises(RuntimeError, match="naive base out of range"):
        fam.asr_batch(T, depth, good["ops"], good["brlen"], good["er"], good["pi"], good["rates"], bad, 1)
    bad_ops = ops.copy()
    bad_ops[0, 1] = 99
    with pytest.raises(RuntimeError, match="malformed schedule op"):
        fam.asr_batch(T, depth, bad_ops[None], good["brlen"], good["er"], good["pi"], good["rates"], good["naive"], 1)
    fam.close()


def test_asr_rejected_device_schedule_gets_the_sentinel():
    """lh_asr_batch_device on a batch whose DEVICE-resident schedules hold one malformed sample (a pop from the wrong stack
    slot: every field in range): that sample's anc and rate_choice rows come back as 0xff in every byte -- states are bytes
    and have no NaN --, lh_family_status reports it, the other samples keep the draws of the clean batch.  (Its own
    process: the device buffers are torch tensors, and torch brings its own HIP runtime, which has to come up first.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "asr_device_worker.py")], capture_output=True, text=True,
                       timeout=300, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["clean_status"] == "" and res["clean_max_state"] <= 3
    assert "malformed schedule" in res["bad_status"]
    assert res["victim_all_ff"] and res["others_unchanged"]


def test_asr_committed_vectors(hip, data_dir):
    """The HIP path against the committed oracle-derived vectors (tests/golden/asr_goldens.json)."""
    import json
    import linearham_amd
    g = json.load(open(os.path.join(os.path.dirname(data_dir), "asr_goldens.json")))
    h = orc.PhyloHMM(os.path.join(data_dir, "phylo_hmm_input.yaml"), 0, os.path.join(data_dir, "hmm_params"), 0)
    fam = linearham_amd.Family(db.build_family_desc(h), hip)
    children, root, brlen = db.tree_arrays(orc.parse_newick(open(os.path.join(data_dir, "newton.tree")).read()),
                                           h.xmsa_labels)
    assert [int(x) for x in children] == g["children"] and int(root) == g["root"]
    ops, depth = hip.schedule_tree(4, children, root)
    n = len(g["samples"])
    naive = np.tile(np.array(g["naive"], dtype=np.uint8), (n, 1))
    rates = np.tile(orc.gamma_rates_mean(1.0, 4), (n, 1))
    anc, choice = fam.asr_batch(4, depth, np.tile(ops, (n, 1, 1)), np.tile(brlen, (n, 1)), [[1.0] * 6] * n,
                                [[0.17, 0.19, 0.25, 0.39]] * n, rates, naive, g["seed"], 0)
    fam.close()
    for i, smp in enumerate(g["samples"]):
        assert choice[i].tolist() == smp["rate_choice"]
        assert anc[i].tolist() == smp["anc"]


def test_asr_full_size_known_answer(hip, tmp_path):
    """BASELINE.json configs[2] shape (100 leaves x 400 sites), 256 tree samples: with every branch at 1e-6 the
    posterior of a constant alignment column is concentrated on that base at every inner node (each node differs
    with probability ~1e-6), so the sampled ancestral states are known without an oracle run; and a second call
    reproduces the first bit for bit."""
    import linearham_amd
    from tools import synth_family as sf
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec(n_samples=8), out)
    h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    fam = linearham_amd.Family(db.build_family_desc(h), hip)
    T, L = h.msa.shape[0] + 1, h.msa.shape[1]
    n = 256
    ops, brl, depth = [], [], 0
    for i in range(n):
        children, root, brlen = db.tree_arrays(orc.parse_newick(rows[i % len(rows)]["tree"]), h.xmsa_labels)
        o, d = hip.schedule_tree(T, children, root)
        ops.append(o)
        brl.append(np.full_like(brlen, 1e-6))
        depth = max(depth, d)
    const = np.all(h.msa == h.msa[0:1], axis=0) & (h.msa[0] < 4)
    assert const.sum() > 50
    naive = np.where(const, h.msa[0], 4).astype(np.uint8)
    er, pi = [[1.0] * 6] * n, [[0.25] * 4] * n
    rates = np.tile(orc.gamma_rates_mean(0.5, 4), (n, 1))
    args = (T, depth, np.stack(ops), np.stack(brl), er, pi, rates, np.tile(naive, (n, 1)), 2026)
    anc, choice = fam.asr_batch(*args)
    anc2, choice2 = fam.asr_batch(*args)
    fam.close()
    assert np.array_equal(anc, anc2) and np.array_equal(choice, choice2)
    want = h.msa[0][const].astype(np.uint8)
    got = anc[:, :, const]                                   # [n][T-2][constant sites]
    wrong = int((got != want[None, None, :]).sum())
    assert wrong <= 5, (wrong, got.size)                     # expectation ~ got.size * 1e-6 * 3 (< 1)
    assert choice.max() < 4 and len(np.unique(choice)) == 4  # all four categories occur over 256 x 400 draws


@pytest.mark.parametrize("seed", [201, 202, 203, 204, 205, 206])
def test_asr_random_small_families(hip, tmp_path, seed):
    """Differently shaped small families (locus, leaves 3..40, allele counts, divergence, R drawn from the seed):
    every one gives K3 another tree shape, pattern set and category split; draws must equal the oracle's."""
    from tools import synth_family as sf
    rng = np.random.default_rng(seed)
    locus = ["igh", "igh", "igk", "igl"][int(rng.integers(4))]
    kw = dict(locus=locus, seed=seed, n_leaves=int(rng.integers(3, 41)), n_samples=3,
              n_v=int(rng.integers(1, 9)), n_j=int(rng.integers(1, 6)), n_nni=int(rng.integers(0, 4)),
              divergence=float(rng.choice([0.0, 0.05, 0.3])))
    if locus == "igh":
        kw["n_d"] = int(rng.integers(1, 6))
    out = str(tmp_path / "fam")
    sf.generate(sf.Spec.small(**kw), out)
    h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    R = int(rng.choice([1, 2, 3, 4, 6]))
    mism, total, anc, choice = _run(hip, h, rows, R, seed=seed, first_sample=int(rng.integers(0, 1 << 40)), rng=rng)
    assert mism == 0, (mism, total)
    assert choice.max() < R

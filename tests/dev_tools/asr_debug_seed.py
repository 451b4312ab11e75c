"""Where the device's ancestral-sequence draws and oracle/asr_oracle.py differ on the families of random_sweep_asr.py: per seed the
sites whose rate category differs (reference / device) and the sites whose states differ under an agreeing category.
usage (GPU box, repo root): python tests/dev_tools/asr_debug_seed.py seed [seed ...]"""
import os, sys, tempfile, shutil
sys.path.insert(0, os.getcwd())
import numpy as np
import linearham_amd
import tests.test_gpu_asr as ta
from oracle import linearham_oracle as orc, asr_oracle as ao
from tools import synth_family as sf
from tests import desc_builder as db
lib = linearham_amd.load_library()
def kw_of(seed):
    rng = np.random.default_rng(seed)
    locus = ["igh", "igh", "igk", "igl"][int(rng.integers(4))]
    kw = dict(locus=locus, seed=seed, n_samples=2, n_leaves=int(rng.integers(3, 70)), n_v=int(rng.integers(1, 5)),
              n_j=int(rng.integers(1, 4)), ragged=int(rng.choice([0, 0, 4, 10])), ambiguous=float(rng.choice([0.0, 0.0, 0.01, 0.05])),
              tree_shape=str(rng.choice(["stepwise", "stepwise", "balanced"])), n_nni=int(rng.integers(0, 4)))
    if locus == "igh": kw["n_d"] = int(rng.integers(1, 4))
    return kw, rng
for seed in map(int, sys.argv[1:]):
    kw, rng = kw_of(seed)
    out = tempfile.mkdtemp(prefix="lh_asrdbg_")
    sf.generate(sf.Spec.small(**kw), out)
    h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    R = int(rng.choice([1, 3, 4, 8]))
    first_sample = int(rng.integers(0, 1000))
    rng2 = rng
    import linearham_amd
    fam = linearham_amd.Family(db.build_family_desc(h), lib)
    T = h.msa.shape[0] + 1; L = h.msa.shape[1]
    ops, brl, trees, depth = [], [], [], 0
    for s in rows:
        children, root, brlen = db.tree_arrays(orc.parse_newick(s["tree"]), h.xmsa_labels)
        o, d = lib.schedule_tree(T, children, root); ops.append(o); brl.append(brlen); trees.append((children, root, brlen)); depth = max(depth, d)
    n = len(rows)
    rates = np.stack([orc.gamma_rates_mean(s["alpha"], R) for s in rows])
    naive = rng2.integers(0, 5, size=(n, L)).astype(np.uint8)
    anc, choice = fam.asr_batch(T, depth, np.stack(ops), np.stack(brl), [s["er"] for s in rows], [s["pi"] for s in rows], rates, naive, seed, first_sample)
    print("seed", seed, "R", R, "T", T, "L", L, "form", fam.k1_form(), "alpha", [s["alpha"] for s in rows], "rates", rates[0])
    fam.close()
    for i, s in enumerate(rows):
        children, root, brlen = trees[i]
        c_ref, a_ref, extra = ao.asr_sample(children, root, brlen, T, h.msa, naive[i], s["er"], np.asarray(s["pi"]), rates[i], seed, first_sample + i)
        dr = np.nonzero(c_ref != choice[i])[0]
        same = c_ref == choice[i]
        ds = np.nonzero((a_ref[:, same] != anc[i][:, same]).any(axis=0))[0]
        print(" sample", i, "rate mismatches at sites", dr.tolist(), "ref", c_ref[dr].tolist(), "got", choice[i][dr].tolist(), "| state mismatches (rate agreeing) at", np.nonzero(same)[0][ds].tolist())
    shutil.rmtree(out, ignore_errors=True)

"""Batch sizes around every boundary of lh_eval_batch (host pointers: staging sub-chunks of 12 288 through two pinned slots, launch
groups of 49 152, launch parts): a small family whose eight distinct tree samples are cycled to n rows; every row's log-likelihood
and rates must carry the bits of its tree sample's in the 8-row call, whatever n (builder-run).
usage (GPU box, repo root): python tests/dev_tools/batch_boundaries.py"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import linearham_amd  # noqa: E402
from oracle import linearham_oracle as orc  # noqa: E402
from tests import desc_builder as db  # noqa: E402
from tools import synth_family as sf  # noqa: E402

lib = linearham_amd.load_library()
out = tempfile.mkdtemp(prefix="lh_bounds_")
bad = 0
try:
    sf.generate(sf.Spec.small(n_leaves=16, n_samples=8, seed=9, ragged=4, ambiguous=0.02), out)
    h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    T = h.msa.shape[0] + 1
    fam = linearham_amd.Family(db.build_family_desc(h), lib)
    ops, brl, depth = [], [], 0
    for s in rows:
        children, root, brlen = db.tree_arrays(orc.parse_newick(s["tree"]), h.xmsa_labels)
        o, d = lib.schedule_tree(T, children, root)
        ops.append(o)
        brl.append(brlen)
        depth = max(depth, d)
    ops, brl = np.stack(ops), np.stack(brl)
    er, pi, al = np.array([s["er"] for s in rows]), np.array([s["pi"] for s in rows]), np.array([s["alpha"] for s in rows])
    ll8, res8 = fam.eval_batch(T, depth, ops, brl, er, pi, al, 4, want=("rates",))
    assert np.all(np.isfinite(ll8))
    sizes = [1, 2, 7, 63, 64, 65, 4095, 4096, 4097, 6143, 6144, 6145, 12287, 12288, 12289, 24575, 24577, 36865, 49151, 49152, 49153,
             61441, 98303, 98305, 100003]
    for n in sizes:
        idx = np.arange(n) % 8
        ll, res = fam.eval_batch(T, depth, ops[idx], brl[idx], er[idx], pi[idx], al[idx], 4, want=("rates",))
        ok = np.array_equal(ll, ll8[idx]) and np.array_equal(res["rates"], res8["rates"][idx])
        bad += 0 if ok else 1
        print("n = %6d: %s" % (n, "identical bits" if ok else "DIFFERENT at rows %s" % np.nonzero(ll != ll8[idx])[0][:8].tolist()), flush=True)
    fam.close()
finally:
    shutil.rmtree(out, ignore_errors=True)
print("batch boundaries: %d of %d sizes differ" % (bad, len(sizes)))
sys.exit(1 if bad else 0)

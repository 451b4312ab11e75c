#!/usr/bin/env python3
"""Oracle-derived vectors for the ancestral-sequence sampling step (NOT reference outputs: phylomd / R are
absent, see oracle/asr_oracle.py).  They freeze the random stream and the draw rule so that oracle and kernel
cannot drift together: tests/test_asr_oracle.py re-derives them on CPU, tests/test_gpu_asr.py checks the
HIP path against the committed file.

    python tests/golden/make_asr_goldens.py        # rewrites tests/golden/asr_goldens.json

Case: the reference's toy family (tests/golden/data/phylo_hmm_input.yaml + newton.tree, copied from the
reference's data/), er = 1, pi = (.17,.19,.25,.39), alpha = 1, R = 4, naive[j] = j mod 5, seed 5, samples 0..3.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def case():
    from oracle import asr_oracle as ao
    from oracle import linearham_oracle as orc
    from tests import desc_builder as db
    d = os.path.join(HERE, "data")
    h = orc.PhyloHMM(os.path.join(d, "phylo_hmm_input.yaml"), 0, os.path.join(d, "hmm_params"), 0)
    tree = orc.parse_newick(open(os.path.join(d, "newton.tree")).read())
    children, root, brlen = db.tree_arrays(tree, h.xmsa_labels)
    er, pi = [1.0] * 6, np.array([0.17, 0.19, 0.25, 0.39])
    rates = orc.gamma_rates_mean(1.0, 4)
    L = h.msa.shape[1]
    naive = (np.arange(L) % 5).astype(np.uint8)
    out = {"_note": "oracle-derived (oracle/asr_oracle.py), not reference output", "seed": 5, "naive": naive.tolist(),
           "children": [int(x) for x in children], "root": int(root), "samples": []}
    for i in range(4):
        choice, anc, _ = ao.asr_sample(children, root, brlen, 4, h.msa, naive, er, pi, rates, 5, i)
        out["samples"].append({"rate_choice": choice.tolist(), "anc": anc.tolist()})
    out["uniform_seed5_sample2_site7_draws0to3"] = [float(ao.uniform(5, 2, np.array([7]), k)[0]) for k in range(4)]
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "asr_goldens.json"), "w") as f:
        json.dump(case(), f, indent=1)
    print("wrote asr_goldens.json")

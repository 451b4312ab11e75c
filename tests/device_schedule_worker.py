"""Helper of tests/test_device_schedules.py (its own process: the K1 form is chosen by environment variables read
once per process).  Evaluates a small synthetic family through lh_eval_batch_device with one sample's DEVICE-RESIDENT
schedule corrupted in a given way; prints a JSON line with what came back."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(mode, n_leaves, repeat=1):
    import numpy as np
    import torch
    import linearham_amd
    from oracle import linearham_oracle as orc
    from tests import desc_builder as db
    from tools import synth_family as sf
    out = "/tmp/lh_devsched_%d_%d" % (n_leaves, os.getpid())
    # small: one one-site wave per rate; from 30 leaves on a 400-site family whose 100+ patterns need two-site waves
    spec = sf.Spec.small(n_leaves=n_leaves, n_samples=6, seed=31) if n_leaves < 30 else \
        sf.Spec(n_leaves=n_leaves, n_sites=400, n_v=24, n_d=6, n_j=4, n_samples=6, seed=31)
    sf.generate(spec, out)
    h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    hip = linearham_amd.load_library()
    fam = linearham_amd.Family(db.build_family_desc(h), hip)
    T = h.msa.shape[0] + 1
    ops, brl, depth = [], [], 0
    for r in rows:
        children, root, brlen = db.tree_arrays(orc.parse_newick(r["tree"]), h.xmsa_labels)
        o, d = hip.schedule_tree(T, children, root)
        ops.append(np.asarray(o, dtype=np.int32).reshape(-1, 4)), brl.append(brlen)
        depth = max(depth, d)
    ops = np.stack(ops)
    if repeat > 1:       # the same samples over and over: a batch large enough for K0c's thread-per-sample form
        ops = np.tile(ops, (repeat, 1, 1))
        brl = brl * repeat
        rows = rows * repeat
    victim = 2
    k_tip = next(k for k in range(ops.shape[1]) if (ops[victim, k, 0] & 15) == 1)     # a tip-into-accumulator op
    if mode == "tip":            # a tip number far outside the alignment: would index the MSA and the tip table out of bounds
        ops[victim, k_tip, 1] = 1 << 20
    elif mode == "kind":         # an op kind that does not exist
        ops[victim, 0, 0] = (ops[victim, 0, 0] & ~15) | 7
    elif mode == "node":         # a branch node beyond 2T - 2: would index the branch lengths out of bounds
        ops[victim, k_tip, 2] = 5 * T
    elif mode == "rank":         # the running matrix count the register-stack form's prologue places matrices by, zeroed
        ops[victim, :, 0] &= 0xff
    elif mode == "slot":         # a stack slot beyond the depth the kernel was built for
        k_pop = next(k for k in range(ops.shape[1]) if (ops[victim, k, 0] & 15) == 2)
        ops[victim, k_pop, 3] = 40
    elif mode in ("popslot", "pushslot", "unbalanced"):
        # Stack discipline broken with every field IN RANGE: nothing would index out of bounds, the kernel would return a
        # finite, wrong likelihood.  popslot: a pop takes another slot than the last one pushed; pushslot: a push goes
        # over a live slot / skips one; unbalanced: the last pop becomes a tip-into-accumulator op (a sibling is left).
        kinds = ops[victim, :, 0] & 15
        pushes = [k for k in range(ops.shape[1]) if ops[victim, k, 0] & 16]
        pops = [k for k in range(ops.shape[1]) if kinds[k] == 2]
        assert pushes and pops, "the victim's tree has no pending sibling at all"
        if mode == "popslot":
            k = pops[0]
            ops[victim, k, 3] = 1 if ops[victim, k, 3] == 0 else 0
        elif mode == "pushslot":
            k = pushes[0]
            ops[victim, k, 3] = 1 if ops[victim, k, 3] == 0 else 0
        else:
            k = pops[-1]
            ops[victim, k, 0] = (ops[victim, k, 0] & ~15) | 1
            ops[victim, k, 1] = 1                      # a tip where the popped node was (fields in range)
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    d_ops, d_brl = t(ops, np.int32), t(np.stack(brl), np.float64)
    d_er, d_pi = t([r["er"] for r in rows], np.float64), t([r["pi"] for r in rows], np.float64)
    d_al = t([r["alpha"] for r in rows], np.float64)
    ll = torch.zeros(len(rows), dtype=torch.float64, device=dev)
    fam.eval_batch_device(len(rows), T, depth, d_ops.data_ptr(), d_brl.data_ptr(), d_er.data_ptr(), d_pi.data_ptr(),
                          d_al.data_ptr(), 4, ll.data_ptr())
    status = ""
    try:
        fam.status()
    except RuntimeError as e:
        status = str(e)
    second = ""
    try:
        fam.status()             # the error state is reported once
    except RuntimeError as e:
        second = str(e)
    got = ll.cpu().numpy()
    # the same corrupted batch through the HOST-pointer entry point: refused before anything reaches the device
    host_error = ""
    try:
        fam.eval_batch(T, depth, ops, np.stack(brl), [r["er"] for r in rows], [r["pi"] for r in rows],
                       [r["alpha"] for r in rows], 4)
    except RuntimeError as e:
        host_error = str(e)
    ll_out = [None if not np.isfinite(x) else float(x) for x in got]
    form = fam.k1_form()
    oracle = []
    if mode == "none":       # the clean run is compared with the oracle by the caller, not only with other runs of itself
        for r in rows[:len(rows) // repeat]:
            h.initialize_phylo_parameters(r["tree"], r["er"], r["pi"], r["alpha"], 4, is_path=False)
            h.initialize_phylo_emission()
            oracle.append(float(h.log_likelihood()))
    if repeat > 1:       # a digest instead of the whole vector: which samples are NaN, and whether the copies agree
        base = len(ll_out) // repeat
        agree = all(ll_out[i] == ll_out[i % base] for i in range(len(ll_out)) if i % base != victim)
        print(json.dumps({"status": status, "second": second, "host_error": host_error, "n": len(ll_out),
                          "nan_at": [i for i, x in enumerate(ll_out) if x is None][:8], "copies_agree": agree,
                          "ll": ll_out[:base], "form": form, "oracle": oracle}))
    else:
        print(json.dumps({"status": status, "second": second, "host_error": host_error, "ll": ll_out, "form": form,
                          "oracle": oracle}))
    fam.close()
    import shutil
    shutil.rmtree(out, ignore_errors=True)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1)

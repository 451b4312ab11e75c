"""Helper of tests/test_gpu_asr.py::test_asr_rejected_device_schedule_gets_the_sentinel (its own process: torch first)."""
import ctypes as C
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    import linearham_amd
    from oracle import linearham_oracle as orc
    from tests import desc_builder as db
    from tools import synth_family as sf
    hip = linearham_amd.load_library()
    out = tempfile.mkdtemp(prefix="lh_asrdev_")
    try:
        sf.generate(sf.Spec.small(n_leaves=12, n_samples=4, seed=31), out)
        h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
        rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    finally:
        shutil.rmtree(out, ignore_errors=True)
    fam = linearham_amd.Family(db.build_family_desc(h), hip)
    T, L, R = h.msa.shape[0] + 1, h.msa.shape[1], 4
    ops, brl, depth = [], [], 0
    for r in rows:
        children, root, brlen = db.tree_arrays(orc.parse_newick(r["tree"]), h.xmsa_labels)
        o, d = hip.schedule_tree(T, children, root)
        ops.append(np.asarray(o, dtype=np.int32).reshape(-1, 4)), brl.append(brlen)
        depth = max(depth, d)
    ops = np.stack(ops)
    n = len(rows)
    rates = np.stack([orc.gamma_rates_mean(r["alpha"], R) for r in rows])
    naive = np.random.default_rng(2).integers(0, 4, size=(n, L)).astype(np.uint8)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)

    def run(ops_arr):
        d_ops, d_brl = t(ops_arr, np.int32), t(np.stack(brl), np.float64)
        d_er, d_pi = t([r["er"] for r in rows], np.float64), t([r["pi"] for r in rows], np.float64)
        d_rates, d_naive = t(rates, np.float64), t(naive, np.uint8)
        anc = torch.full((n, T - 2, L), 0x55, dtype=torch.uint8, device=dev)      # stale bytes that must not survive
        choice = torch.full((n, L), 0x55, dtype=torch.uint8, device=dev)
        hip.check(hip.lib.lh_asr_batch_device(fam.handle, n, T, depth, d_ops.data_ptr(), d_brl.data_ptr(), d_er.data_ptr(),
                                              d_pi.data_ptr(), d_rates.data_ptr(), R, d_naive.data_ptr(), C.c_uint64(9),
                                              C.c_uint64(0), anc.data_ptr(), choice.data_ptr(), None))
        status = ""
        try:
            fam.status()
        except RuntimeError as e:
            status = str(e)
        return anc.cpu().numpy(), choice.cpu().numpy(), status
    anc0, choice0, status0 = run(ops)
    bad = ops.copy()
    victim = 1
    pops = [k for k in range(bad.shape[1]) if (bad[victim, k, 0] & 15) == 2]
    assert pops, "the victim's tree has a pending sibling"
    bad[victim, pops[0], 3] = 1 if bad[victim, pops[0], 3] == 0 else 0
    anc1, choice1, status1 = run(bad)
    others = all(np.array_equal(anc1[i], anc0[i]) and np.array_equal(choice1[i], choice0[i]) for i in range(n) if i != victim)
    print(json.dumps({"clean_status": status0, "clean_max_state": int(max(anc0.max(), choice0.max())), "bad_status": status1,
                      "victim_all_ff": bool((anc1[victim] == 0xff).all() and (choice1[victim] == 0xff).all()),
                      "others_unchanged": bool(others)}))
    fam.close()


if __name__ == "__main__":
    main()

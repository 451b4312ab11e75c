"""Debug helper (GPU box): evaluates a synthetic family with the assembly walk and with the C++ walk
(LH_K1_CXX_WALK=1, separate processes) and prints where their per-column emissions differ.
usage: python tests/dev_tools/debug_walk.py [n_leaves] [seed]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(n_leaves, seed):
    import numpy as np
    import linearham_amd
    from oracle import linearham_oracle as orc
    from tests import desc_builder as db
    from tools import synth_family as sf
    n_sites = int(os.environ.get("LH_DBG_SITES", "400"))
    out = "/tmp/dbg_fam_%d_%d_%d" % (n_leaves, seed, n_sites)
    if not os.path.exists(out + "/meta.json"):
        sf.generate(sf.Spec(n_leaves=n_leaves, n_sites=n_sites, n_v=24, n_d=6, n_j=4, n_samples=4, seed=seed), out)
    h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
    rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
    hip = linearham_amd.load_library()
    fam = linearham_amd.Family(db.build_family_desc(h), hip)
    T = h.msa.shape[0] + 1
    ops, brl, depth = [], [], 0
    for s in rows:
        children, root, brlen = db.tree_arrays(orc.parse_newick(s["tree"]), h.xmsa_labels)
        o, d = hip.schedule_tree(T, children, root)
        ops.append(o), brl.append(brlen)
        depth = max(depth, d)
    ll, res = fam.eval_batch(T, depth, np.stack(ops), np.stack(brl), [s["er"] for s in rows], [s["pi"] for s in rows],
                             [s["alpha"] for s in rows], 4, want=("xmsa_emission",))
    descs = []
    if os.environ.get("LH_DBG_DESC"):   # the rewritten walk of sample 0, emulated from the schedule (K0c's rules)
        o = np.asarray(ops[0]).reshape(-1, 4)
        k = 0
        while k < len(o):
            x, y, z, w = [int(v) for v in o[k]]
            kd, push = x & 15, bool(x & 16)
            nk = int(o[k + 1][0]) & 15 if k + 1 < len(o) else -1
            if kd == 0 and nk == 2 and push:
                descs.append("ctab(%d,%d)" % (y, z)); k += 2
            elif kd == 0 and nk == 1:
                descs.append("ctip(%d,%d;%d)%s" % (y, z, int(o[k + 1][1]), " push" if push else "")); k += 2
            elif kd == 0:
                descs.append("cherry(%d,%d)%s" % (y, z, " push" if push else "")); k += 1
            elif kd == 1:
                descs.append("tip(%d)" % y); k += 1
            else:
                descs.append("pop"); k += 1
    import ctypes as C
    npat, nu = C.c_int32(), C.c_int32()
    hip.lib.lh_family_info(fam.handle, C.byref(npat), C.byref(nu))
    print(json.dumps({"n_pat": npat.value, "depth": depth, "ll": [float(x) for x in ll], "em": res["xmsa_emission"].tolist(), "walk": descs}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    import numpy as np
    n_leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 123
    r = {}
    for mode, env in (("asm", {}), ("cxx", {"LH_K1_CXX_WALK": "1"})):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n_leaves), str(seed)],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        if p.returncode:
            print(mode, "failed:", p.stderr[-2000:])
            sys.exit(1)
        r[mode] = json.loads(p.stdout.strip().splitlines()[-1])
    print("n_pat", r["asm"].get("n_pat"), "depth", r["asm"]["depth"], "ll asm", r["asm"]["ll"], "ll cxx", r["cxx"]["ll"])
    if r["asm"]["walk"]:
        print("walk of sample 0:", " | ".join("%d:%s" % (i, d) for i, d in enumerate(r["asm"]["walk"])))
    a, c = np.array(r["asm"]["em"]), np.array(r["cxx"]["em"])
    for i in range(a.shape[0]):
        bad = np.where(~np.isclose(a[i], c[i], rtol=1e-12, atol=0, equal_nan=False))[0]
        print("sample", i, "columns differing:", len(bad), "of", a.shape[1], "first", bad[:20].tolist(),
              "asm", a[i][bad[:4]].tolist(), "cxx", c[i][bad[:4]].tolist())

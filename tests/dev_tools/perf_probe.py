#!/usr/bin/env python3
"""Scratch performance probe (NOT bench.py): times the HIP kernels on a synthetic family using the
test-infrastructure descriptor builder.  Usage: python tests/dev_tools/perf_probe.py [preset] [n_samples] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import linearham_amd
from oracle import linearham_oracle as orc
from tests import desc_builder as db
from tools import synth_family as sf

preset = sys.argv[1] if len(sys.argv) > 1 else "config2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
out = "/tmp/lh_probe_" + preset
spec = {"config2": sf.Spec(n_samples=64), "small": sf.Spec.small(), "config4": sf.Spec(n_leaves=500, n_sites=600, n_samples=32)}[preset]
t0 = time.time()
if not os.path.exists(out + "/trees.tsv"):
    sf.generate(spec, out)
print("generate %.1fs" % (time.time() - t0), flush=True)
t0 = time.time()
h = orc.PhyloHMM(out + "/cluster.yaml", 0, out + "/hmm_params", 0)
print("oracle HMM %.1fs; C=%d S_vd=%d S_dj=%d" % (time.time() - t0, h.xmsa.shape[1], len(h.vd_junction.state_strs), len(h.dj_junction.state_strs)), flush=True)
desc = db.build_family_desc(h)
G = sum(len(s.xmsa_inds) for s in (desc.vpadding, desc.vgerm, desc.dgerm, desc.jgerm, desc.jpadding))
print("G=%d W_vd=%d W_dj=%d" % (G, desc.vd.n_rows, desc.dj.n_rows))
lib = linearham_amd.load_library()
fam = linearham_amd.Family(desc, lib)
rows = sf.read_trees_tsv(out + "/trees.tsv")
T = h.msa.shape[0] + 1
ops, brl, depth = [], [], 0
t0 = time.time()
for r in rows:
    tree = orc.parse_newick(r["tree"])
    c, root, b = db.tree_arrays(tree, h.xmsa_labels)
    o, d = lib.schedule_tree(T, c, root)
    ops.append(o); brl.append(b); depth = max(depth, d)
print("flatten %d trees %.2fs, max depth %d" % (len(rows), time.time() - t0, depth), flush=True)
idx = np.arange(n) % len(rows)
ops = np.stack(ops)[idx]; brl = np.stack(brl)[idx]
er = np.array([r["er"] for r in rows])[idx]; pi = np.array([r["pi"] for r in rows])[idx]
alpha = np.array([r["alpha"] for r in rows])[idx]
dev = torch.device("cuda:0")
d_ops = torch.from_numpy(ops).to(dev); d_brl = torch.from_numpy(brl).to(dev)
d_er = torch.from_numpy(er).to(dev); d_pi = torch.from_numpy(pi).to(dev); d_alpha = torch.from_numpy(alpha).to(dev)
d_ll = torch.zeros(n, dtype=torch.float64, device=dev)
def step():
    fam.eval_batch_device(n, T, depth, d_ops.data_ptr(), d_brl.data_ptr(), d_er.data_ptr(), d_pi.data_ptr(), d_alpha.data_ptr(), 4, d_ll.data_ptr(), 0)
step(); torch.cuda.synchronize()
fam.profile_enable(True)
t0 = time.time()
for _ in range(reps):
    step()
torch.cuda.synchronize()
dt = (time.time() - t0) / reps
p = fam.profile_read()
print("n=%d  %.3f ms/step  %.0f evals/s" % (n, dt * 1e3, n / dt))
print({k: (v / reps if k.endswith("ms") else v) for k, v in p.items()})
ll = d_ll.cpu().numpy()
print("loglik[:4]", ll[:4])
# parity of the first 2 samples against the dense oracle
for i in range(2):
    r = rows[i]
    h.initialize_phylo_parameters(r["tree"], r["er"], r["pi"], r["alpha"], 4, is_path=False)
    h.initialize_phylo_emission()
    ref = h.log_likelihood()
    print("sample %d gpu %.10f oracle %.10f rel %.2e" % (i, ll[i], ref, abs(ll[i] - ref) / abs(ref)))

import os, sys, tempfile, numpy as np
sys.path.insert(0, os.getcwd())
from tools import synth_family as sf
from oracle import linearham_oracle as orc, oracle_c
from tests import desc_builder as db
import linearham_amd
from linearham_amd import host
spec = sf.Spec(n_leaves=500, n_sites=600, n_samples=64)
out = tempfile.mkdtemp()
sf.generate(spec, out)
hmm = host.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
flat = hmm.flatten_tsv(os.path.join(out, "trees.tsv"), 64)
lib = linearham_amd.load_library()
import ctypes as C
ll = np.zeros(64)
p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
lib.check(lib.lib.lh_eval_batch(C.c_void_p(flat["family"]), 64, flat["n_tips"], flat["max_depth"],
          p(np.ascontiguousarray(flat["ops"]), C.c_int32), p(np.ascontiguousarray(flat["brlen"]), C.c_double),
          p(flat["er"], C.c_double), p(flat["pi"], C.c_double), p(flat["alpha"], C.c_double), 4, p(ll, C.c_double), None))
bad=[i for i in range(64) if not np.isfinite(ll[i])]
print("gpu non-finite", bad, ll[bad])
oracle_c.build()
h = orc.PhyloHMM(os.path.join(out, "cluster.yaml"), 0, os.path.join(out, "hmm_params"), 0)
fam = oracle_c.COracleFamily(h, 4)
rows = sf.read_trees_tsv(os.path.join(out, "trees.tsv"))
idx = (bad + [0, 1])[:6]
trees = [db.tree_arrays(orc.parse_newick(rows[i]["tree"]), h.xmsa_labels) for i in idx]
ref = fam.eval(trees, [rows[i]["er"] for i in idx], [rows[i]["pi"] for i in idx], [rows[i]["alpha"] for i in idx], n_threads=6)
print("idx", idx); print("cpu", ref); print("gpu", ll[idx])

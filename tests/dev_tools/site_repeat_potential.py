#!/usr/bin/env python3
"""K1 experiment (ii) of round 2: how much Felsenstein-pruning work could "site repeats" save on the configs[2]
family?  For tree samples of the synthetic RevBayes table it counts (a) the (inner node, site pattern) pairs
whose subtree holds more than one base ("mixed": the only CLVs that differ from a constant-subtree CLV), and
(b) the schedule ops a WAVE of 64 / 128 patterns could skip if it skipped every subtree that is constant for
all of its patterns (in alignment order and with patterns clustered by where their minority tips sit in a
guide tree).  Result (profiles/r02_site_repeat_potential.txt): 35 % of the pairs are mixed -- a 2.9x saving
exists per lane -- but a wave still has to execute 88-97 % of the ops, so the saving is out of reach of a
walk that keeps CLVs in registers with one pattern per lane.
usage: python tests/dev_tools/site_repeat_potential.py [family dir made by tools/synth_family.py]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tools import synth_family as sf
from oracle import linearham_oracle as orc
from tests import desc_builder as db
d = sys.argv[1] if len(sys.argv) > 1 else '/tmp/lh_feed'
if not os.path.exists(d + '/meta.json'):
    sf.generate(sf.Spec(n_samples=64), d)
o = orc.PhyloHMM(d+'/cluster.yaml', 0, d+'/hmm_params', 0)
msa = o.msa   # [n_seqs][L]
n, L = msa.shape
# patterns
cols = {}
for j in range(L):
    cols.setdefault(tuple(msa[:, j]), len(cols))
pats = np.array(list(cols.keys())).T   # [n][n_pat]
pats = pats[:, [k for k in range(pats.shape[1]) if not np.all(pats[:,k]==4)]]
npat = pats.shape[1]
print('patterns', npat, 'constant patterns', sum(len(set(pats[:,k]))==1 for k in range(npat)))
rows = sf.read_trees_tsv(d+'/trees.tsv', max_rows=40)
T = n+1
def subtree_tips(children, root):
    # returns list of (node, tips under node) for inner nodes in post-order
    out = {}
    def rec(v):
        if v < T: return [v]
        a, b = children[2*(v-T)], children[2*(v-T)+1]
        t = rec(a) + rec(b)
        out[v] = t
        return t
    sys.setrecursionlimit(10000)
    rec(root)
    return out
def dfs_tip_order(children, root):
    order = []
    def rec(v):
        if v < T: order.append(v); return
        rec(children[2*(v-T)]); rec(children[2*(v-T)+1])
    rec(root); return order
trees = [db.tree_arrays(orc.parse_newick(r['tree']), o.xmsa_labels) for r in rows]
# guide order from tree 0
order0 = [t for t in dfs_tip_order(trees[0][0], trees[0][1]) if t >= 1]
pos = {t: i for i, t in enumerate(order0)}
# pattern sort key: positions (in guide DFS order) of tips that differ from the pattern's majority base
keys = []
for k in range(npat):
    col = pats[:, k]
    vals, cnt = np.unique(col, return_counts=True)
    maj = vals[np.argmax(cnt)]
    minority = sorted(pos[i+1] for i in range(n) if col[i] != maj)
    keys.append((minority[0] if minority else -1, minority[-1] if minority else -1, len(minority)))
perm = sorted(range(npat), key=lambda k: keys[k])
for label, p in (('as is', list(range(npat))), ('clustered', perm)):
    for wsz in (64, 128):
        groups = [p[i:i+wsz] for i in range(0, npat, wsz)]
        tot_exec = 0; tot = 0; tot_loadk = 0
        for children, root, brlen in trees:
            st = subtree_tips(children, root)
            for g in groups:
                sub = pats[:, g]
                # node is wave-constant if for every pattern in g all tips under node share a state
                const = {}
                for v, tips in st.items():
                    rows_ = sub[[t-1 for t in tips if t >= 1], :]
                    const[v] = bool(np.all(rows_ == rows_[0:1, :]))
                # executed ops: nodes not constant; loadk: constant nodes whose parent is not constant (maximal)
                ex = sum(1 for v in st if not const[v])
                par = {}
                for v in st:
                    for c in (children[2*(v-T)], children[2*(v-T)+1]):
                        par[c] = v
                lk = sum(1 for v in st if const[v] and (v == root or not const[par[v]]))
                tot_exec += ex; tot += len(st); tot_loadk += lk
        print('%-10s wave of %3d patterns: executes %.1f%% of the ops, + %.1f%% table loads' % (label, wsz, 100.0*tot_exec/tot, 100.0*tot_loadk/tot))
# lower bound: per-pattern mixed nodes
tot=0; mixed=0
for children, root, brlen in trees[:10]:
    st = subtree_tips(children, root)
    for v, tips in st.items():
        rows_ = pats[[t-1 for t in tips if t >= 1], :]
        mixed += int(np.sum(np.any(rows_ != rows_[0:1, :], axis=0))); tot += npat
print('per-pattern: %.1f%% of (node, pattern) pairs are mixed' % (100.0*mixed/tot))

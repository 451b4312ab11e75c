#!/usr/bin/env python3
"""Deterministic synthetic clonal families in the reference's own input formats.

Writes, into one directory:
  hmm_params/IGH{V,D,J}_synNN_star_MM.yaml   partis per-allele HMM files (schema of src/Germline.cpp,
                                              src/NTInsertion.cpp, src/NPadding.cpp asserts)
  cluster.yaml                                partis cluster file with linearham-info (src/HMM.cpp:30-44)
  trees.tsv                                   RevBayes .trees table (src/PhyloHMM.cpp:396-400)
so that the same files drive the oracle, the C++ host + HIP path, and bench.py.

Priors follow templates/revbayes_template.rev: er ~ Dirichlet(1^6), pi ~ Dirichlet(1^4),
alpha ~ Exp(1) (floored at 0.05), branch lengths ~ Exp(rate 100) floored at 1e-6, uniform unrooted
topology with `naive` as a tip.  Tree samples = truth tree + random NNI moves + log-normal branch
length jitter.
"""
import argparse
import json
import math
import os

import numpy as np

BASES = "ACGT"


class Spec:
    """Sizes of one synthetic family.  Defaults = BASELINE.json configs[2] (100 leaves x 400 sites,
    full V/D/J germline set)."""

    def __init__(self, n_leaves=100, n_sites=400, n_v=200, n_d=30, n_j=12, len_v=296, len_d=(11, 37),
                 len_j=(48, 63), v_l_width=3, v_r_width=10, n_samples=256, n_nni=4, seed=20261004,
                 v_ancestors=7, d_ancestors=4, j_ancestors=3, divergence=0.05, locus="igh", brlen_mean=0.01,
                 ragged=0, ambiguous=0.0, tree_shape="stepwise", shm_indels=0):
        # ragged: reads of unequal extent -- every sequence gets its first a and last b sites replaced by N, a and b
        #   uniform in [0, ragged] (partis pads such reads with N: src/HMM.cpp:69-83 takes them as they come and libpll
        #   encodes N as 1111, src/PhyloHMM.cpp:368-370); ambiguous: fraction of the remaining cells set to N.
        # tree_shape: "stepwise" (uniform stepwise addition: ladder-like, schedule stack depth 3-4), "balanced" (a
        #   perfectly balanced tree over the leaves with naive on a branch of its own: stack depth ~log2(n_leaves)).
        # shm_indels: the first k sequences are flagged has_shm_indels -- their aligned form goes to indel_reversed_seqs and
        #   input_seqs holds the read as sequenced (one base missing: not alignable as it stands), src/HMM.cpp:74-79.
        self.__dict__.update(locals())
        del self.__dict__["self"]
        v_r = (len_v - v_r_width, len_v)
        d_l = (v_r[1] + 3, v_r[1] + 13) if len_v >= 100 else (v_r[1] + 1, v_r[1] + 5)
        d_r = (d_l[1] + 3, d_l[1] + 15) if len_v >= 100 else (d_l[1] + 2, d_l[1] + 6)
        j_l = (d_r[1] + 4, d_r[1] + 18) if len_v >= 100 else (d_r[1] + 2, d_r[1] + 7)
        if locus == "igh":
            self.flexbounds = {"v_l": (0, v_l_width), "v_r": v_r, "d_l": d_l, "d_r": d_r, "j_l": j_l,
                               "j_r": (n_sites, n_sites)}
        else:  # igk / igl: no D segment, a single V-J junction (src/HMM.cpp:124-131,163-170)
            assert locus in ("igk", "igl")
            j_l = (v_r[1] + 3, v_r[1] + 13) if len_v >= 100 else (v_r[1] + 1, v_r[1] + 5)
            self.flexbounds = {"v_l": (0, v_l_width), "v_r": v_r, "j_l": j_l, "j_r": (n_sites, n_sites)}
            self.n_d = 0
        assert j_l[1] + 4 < n_sites, "n_sites too small for this layout"

    @staticmethod
    def small(seed=7, **kw):
        """A family the dense oracle evaluates in well under a second."""
        d = dict(n_leaves=8, n_sites=62, n_v=4, n_d=3, n_j=3, len_v=20, len_d=(6, 14), len_j=(15, 22),
                 v_l_width=2, v_r_width=5, n_samples=5, n_nni=2, seed=seed, v_ancestors=2,
                 d_ancestors=2, j_ancestors=2, divergence=0.15)
        d.update(kw)
        return Spec(**d)


# --------------------------------------------------------------------------------------------------
# germline set
# --------------------------------------------------------------------------------------------------

def _norm_map(keys, weights):
    w = np.asarray(weights, dtype=float)
    w = np.round(w / w.sum(), 6)
    w[np.argmax(w)] += round(1.0 - w.sum(), 6)
    return {k: float(round(x, 6)) for k, x in zip(keys, w)}


def _fmt_map(m):
    return "{" + ", ".join("%s: %s" % (k, repr(float(v))) for k, v in m.items()) + "}"


def _emission(base, mu):
    return {b: (round(1.0 - 3 * mu, 6) if b == base else mu) for b in BASES}


def _erosion_profile(n, width, rng):
    """end-probabilities for the last `width` positions (last position = 1)."""
    out = np.zeros(n)
    w = min(width, n)
    prof = np.sort(rng.uniform(0.02, 0.5, size=w))
    out[n - w:] = np.round(prof, 4)
    out[-1] = 1.0
    return out


def write_allele(path, name, gtype, seq, gene_prob, rng):
    n = len(seq)
    lines = ["extras: {gene_prob: %s}" % repr(float(gene_prob)), "name: %s" % name, "states:"]

    def state(nm, emissions, extras, transitions):
        lines.append("- emissions: null" if emissions is None else "- emissions:")
        if emissions is not None:
            lines.append("    probs: " + _fmt_map(emissions))
            lines.append("    track: nukes")
        lines.append("  extras: " + extras)
        lines.append("  name: " + nm)
        lines.append("  transitions: " + _fmt_map(transitions))

    gname = lambda k: "%s_%d" % (name, k)
    end_probs = _erosion_profile(n, 12 if n > 40 else max(2, n // 3), rng)
    mus = np.round(rng.uniform(0.005, 0.06, size=n), 4)
    if gtype == "V":
        pn = round(float(rng.uniform(0.05, 0.4)), 3)
        init = {gname(0): round(1 - pn, 6), "insert_left_N": pn}
        state("init", None, "{}", init)
        state("insert_left_N", {b: 0.25 for b in BASES}, "{ambiguous_emission_prob: 0.25, germline: N}", init)
    else:
        n_in = min(n - 1, 6 if n > 10 else 3)
        keys = [gname(k) for k in range(n_in)] + ["insert_left_" + b for b in BASES]
        w = np.concatenate([np.sort(rng.uniform(0.02, 1.0, size=n_in))[::-1], rng.uniform(0.2, 1.0, size=4)])
        init = _norm_map(keys, w)
        state("init", None, "{}", init)
        for b in BASES:
            w = np.concatenate([np.sort(rng.uniform(0.02, 1.0, size=n_in))[::-1],
                                rng.uniform(0.1, 0.6, size=4)])
            em = {c: (0.94 if c == b else 0.02) for c in BASES}
            state("insert_left_" + b, em, "{germline: %s}" % b, _norm_map(keys, w))
    for k in range(n):
        last = k == n - 1
        if gtype == "J":
            # J genes erode only on the 5' side; the last state hands over to insert_right_N
            if last:
                pe = round(float(rng.uniform(0.02, 0.1)), 3)
                tr = {"end": pe, "insert_right_N": round(1 - pe, 6)}
            else:
                tr = {gname(k + 1): 1.0}
        else:
            e = float(end_probs[k])
            if last:
                tr = {"end": 1.0}
            elif e > 0:
                tr = {gname(k + 1): round(1 - e, 6), "end": e}
            else:
                tr = {gname(k + 1): 1.0}
        state(gname(k), _emission(seq[k], float(mus[k])), "{germline: %s}" % seq[k], tr)
    if gtype == "J":
        state("insert_right_N", {b: 0.25 for b in BASES}, "{ambiguous_emission_prob: 0.25, germline: N}", tr)
    lines.append("tracks:")
    lines.append("  nukes: [A, C, G, T]")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def _mutate(seq, frac, rng):
    s = list(seq)
    for i in range(len(s)):
        if rng.random() < frac:
            s[i] = BASES[(BASES.index(s[i]) + int(rng.integers(1, 4))) % 4]
    return "".join(s)


def make_germline_set(spec, outdir, rng):
    """Returns {"V"|"D"|"J": [(gene name, sequence, relpos)]}.

    Every allele is cut out of one site-aligned root sequence per segment (so that all alleles agree
    on which germline base sits at which MSA site, as Smith-Waterman candidates of ONE rearrangement
    do) and then mutated in two levels: ancestor = <= `divergence` mutant of the root, allele =
    <= `divergence` mutant of its ancestor.  Unrelated or misaligned alleles would differ by more than
    2^1024 in likelihood and trip the reference's 2^(256*d) equalisation overflow
    (src/PhyloHMM.cpp:190-192), giving NaN on both sides."""
    os.makedirs(outdir, exist_ok=True)
    fb = spec.flexbounds
    L = spec.n_sites
    genes = {"V": [], "D": [], "J": []}
    for gtype, n, n_anc in (("V", spec.n_v, spec.v_ancestors), ("D", spec.n_d, spec.d_ancestors),
                            ("J", spec.n_j, spec.j_ancestors)):
        root = "".join(rng.choice(list(BASES), size=L + 80))        # root[s + 40] = base at MSA site s
        ancestors = [root] + [_mutate(root, spec.divergence, rng) for _ in range(n_anc - 1)]
        probs = np.round(rng.dirichlet(np.ones(n)), 6)
        for k in range(n):
            anc = k % n_anc
            if gtype == "V":
                ln = spec.len_v
                relpos = 1 if k == 0 else int(rng.integers(0, fb["v_l"][1] + 1))
            elif gtype == "D":
                ln = int(rng.integers(spec.len_d[0], spec.len_d[1] + 1))
                lo, hi = fb["d_r"][0] - ln, fb["d_l"][1]            # must cover [d_l.second, d_r.first)
                relpos = int(rng.integers(max(lo, fb["v_r"][0] - 5), hi + 1))
                if k == 0:
                    relpos = min(max(fb["d_l"][0] + 1, lo), hi)
            else:
                ln = int(rng.integers(spec.len_j[0], spec.len_j[1] + 1))
                relpos = fb["j_l"][0] + 1 if k == 0 else int(rng.integers(fb["j_l"][0], fb["j_l"][1] + 1))
            seq = ancestors[anc][relpos + 40:relpos + 40 + ln]
            if k >= n_anc:
                seq = _mutate(seq, spec.divergence, rng)
            name = "IG%s%s_syn%d_star_%02d" % (spec.locus[2].upper(), gtype, anc + 1, k // n_anc + 1)
            write_allele(os.path.join(outdir, name + ".yaml"), name, gtype, seq, max(float(probs[k]), 1e-6),
                         rng)
            genes[gtype].append((name.replace("_star_", "*"), seq, relpos))
    return genes


# --------------------------------------------------------------------------------------------------
# trees and sequence evolution
# --------------------------------------------------------------------------------------------------

def random_unrooted_tree(n_tips, rng, brlen_mean=0.01):
    """Uniform stepwise addition.  Returns edges as dict edge_id -> [a, b, length]; tips 0..n-1."""
    edges = {0: [0, n_tips, 0.0], 1: [1, n_tips, 0.0], 2: [2, n_tips, 0.0]}
    nxt_node, nxt_edge = n_tips + 1, 3
    for tip in range(3, n_tips):
        e = int(rng.choice(list(edges.keys())))
        a, b, _ = edges[e]
        edges[e] = [a, nxt_node, 0.0]
        edges[nxt_edge] = [nxt_node, b, 0.0]
        edges[nxt_edge + 1] = [tip, nxt_node, 0.0]
        nxt_node += 1
        nxt_edge += 2
    for e in edges.values():
        e[2] = max(float(rng.exponential(brlen_mean)), 1e-6)
    return edges


def balanced_unrooted_tree(n_tips, rng, brlen_mean=0.01):
    """Tip 0 (naive) and two perfectly balanced subtrees over tips 1..n-1 around one inner node: the shape whose
    post-order schedule needs the deepest stack for its size (~log2 n pending siblings).  Same edge dict as above."""
    edges, nxt = {}, [n_tips]

    def build(tips):
        if len(tips) == 1:
            return tips[0]
        node = nxt[0]
        nxt[0] += 1
        h = len(tips) // 2
        for kid in (build(tips[:h]), build(tips[h:])):
            edges[len(edges)] = [kid, node, 0.0]
        return node

    rest = list(range(1, n_tips))
    centre = nxt[0]
    nxt[0] += 1
    h = len(rest) // 2
    for kid in (0, build(rest[:h]), build(rest[h:])):
        edges[len(edges)] = [kid, centre, 0.0]
    for e in edges.values():
        e[2] = max(float(rng.exponential(brlen_mean)), 1e-6)
    return edges


def _adjacency(edges):
    adj = {}
    for a, b, l in edges.values():
        adj.setdefault(a, []).append((b, l))
        adj.setdefault(b, []).append((a, l))
    return adj


def nni(edges, n_tips, rng):
    """One random nearest-neighbour interchange on an internal edge."""
    internal = [k for k, (a, b, _) in edges.items() if a >= n_tips and b >= n_tips]
    if not internal:
        return
    k = internal[int(rng.integers(len(internal)))]
    a, b, _ = edges[k]
    ea = [j for j, (x, y, _) in edges.items() if j != k and (x == a or y == a)]
    eb = [j for j, (x, y, _) in edges.items() if j != k and (x == b or y == b)]
    ja, jb = ea[int(rng.integers(2))], eb[int(rng.integers(2))]
    for j, old, new in ((ja, a, b), (jb, b, a)):
        x, y, l = edges[j]
        edges[j] = [new if x == old else x, new if y == old else y, l]


def newick(edges, n_tips, labels, rng=None, annotate=True):
    """Unrooted Newick with a trifurcation at the inner node adjacent to tip 1 (arbitrary)."""
    adj = _adjacency(edges)
    root = adj[1][0][0]
    counter = [0]

    def fmt(l):
        return "%.10g" % l

    def rec(node, par):
        counter[0] += 1
        tag = "[&index=%d]" % counter[0] if annotate else ""
        if node < n_tips:
            return labels[node] + tag
        kids = [(nb, l) for nb, l in adj[node] if nb != par]
        return "(" + ",".join(rec(nb, node) + ":" + fmt(l) for nb, l in kids) + ")" + tag

    import sys
    sys.setrecursionlimit(100000)
    return rec(root, -1) + ";"


def gtr_p(er, pi, t):
    S = np.zeros((4, 4))
    k = 0
    for i in range(4):
        for j in range(i + 1, 4):
            S[i, j] = S[j, i] = er[k]
            k += 1
    Q = S * np.asarray(pi)[None, :]
    np.fill_diagonal(Q, 0)
    np.fill_diagonal(Q, -Q.sum(1))
    Q /= -np.sum(np.asarray(pi) * np.diag(Q))
    lam, V = np.linalg.eig(Q)
    return np.real(V @ np.diag(np.exp(lam * t)) @ np.linalg.inv(V))


def evolve(edges, n_tips, root_tip, root_seq, er, pi, site_rates, rng):
    """Evolve root_seq (ints 0..3, -1 = N kept as is) from tip `root_tip` through the tree."""
    adj = _adjacency(edges)
    L = len(root_seq)
    seqs = {root_tip: np.array(root_seq)}
    stack = [(root_tip, -1)]
    urates = np.unique(site_rates)
    while stack:
        node, par = stack.pop()
        for nb, l in adj[node]:
            if nb == par:
                continue
            child = seqs[node].copy()
            for r in urates:
                P = gtr_p(er, pi, l * r)
                P = np.clip(P, 0, None)
                P /= P.sum(1, keepdims=True)
                idx = np.where((site_rates == r) & (seqs[node] >= 0))[0]
                cdf = np.cumsum(P[seqs[node][idx]], axis=1)
                u = rng.random(len(idx))
                child[idx] = (u[:, None] > cdf).sum(1).clip(0, 3)
            seqs[nb] = child
            stack.append((nb, node))
    return seqs


# --------------------------------------------------------------------------------------------------
# family
# --------------------------------------------------------------------------------------------------

def generate(spec, outdir):
    rng = np.random.default_rng(spec.seed)
    os.makedirs(outdir, exist_ok=True)
    genes = make_germline_set(spec, os.path.join(outdir, "hmm_params"), np.random.default_rng(spec.seed + 2))
    fb = spec.flexbounds
    L = spec.n_sites
    # true rearrangement: allele 0 of each segment, placed at its own relpos
    vname, vseq, v_relpos = genes["V"][0]
    jname, jseq, j_relpos = genes["J"][0]
    if spec.locus == "igh":
        dname, dseq, d_relpos = genes["D"][0]
    else:
        dseq, d_relpos = "", fb["v_r"][1]
    rng_s = np.random.default_rng(spec.seed + 3)
    v_end = fb["v_r"][1] - 2                                # V 3' deletion
    d_del5 = 1
    d_start = d_relpos + d_del5
    d_end = min(d_relpos + len(dseq), fb["d_r"][1] - 1) if spec.locus == "igh" else d_start
    j_del5 = 2
    j_start = j_relpos + j_del5
    j_end = min(j_relpos + len(jseq), L)
    naive = np.full(L, -1, dtype=np.int64)
    rnd = lambda n: rng_s.integers(0, 4, size=n)
    naive[v_relpos:v_end] = [BASES.index(c) for c in vseq[:v_end - v_relpos]]
    naive[v_end:d_start] = rnd(d_start - v_end)
    naive[d_start:d_end] = [BASES.index(c) for c in dseq[d_del5:d_del5 + d_end - d_start]]
    naive[d_end:j_start] = rnd(j_start - d_end)
    naive[j_start:j_end] = [BASES.index(c) for c in jseq[j_del5:j_del5 + j_end - j_start]]
    # truth tree over naive (tip 0) + leaves
    T = spec.n_leaves + 1
    labels = ["naive"] + ["s%d" % i for i in range(spec.n_leaves)]
    tree_rng = np.random.default_rng(spec.seed)
    if spec.tree_shape == "balanced":
        edges = balanced_unrooted_tree(T, tree_rng, spec.brlen_mean)
    else:
        edges = random_unrooted_tree(T, tree_rng, spec.brlen_mean)
    er0 = rng_s.dirichlet(np.ones(6))
    pi0 = rng_s.dirichlet(np.ones(4) * 5)
    cat_rates = np.array([0.136954, 0.476752, 1.0, 2.386294])
    site_rates = cat_rates[rng_s.integers(0, 4, size=L)]
    seqs = evolve(edges, T, 0, naive, er0, pi0, site_rates, rng_s)
    to_str = lambda a: "".join("N" if x < 0 else BASES[x] for x in a)
    if spec.ragged or spec.ambiguous:
        # N inside alignment columns (the leaves only; naive_seq stays as the rearrangement made it)
        mrng = np.random.default_rng(spec.seed + 4)
        L_used = int(np.max(np.where(naive >= 0)[0])) + 1   # sites behind it are all-N padding already
        for i in range(1, T):
            s_i = seqs[i] = seqs[i].copy()
            if spec.ragged:
                a, b = (int(x) for x in mrng.integers(0, spec.ragged + 1, size=2))
                s_i[:a] = -1
                s_i[max(L_used - b, 0):L_used] = -1
            if spec.ambiguous:
                s_i[mrng.random(L) < spec.ambiguous] = -1
    # relpos per allele, consistent with the constraints of SURVEY.md 8.1
    relpos = {}
    for name, seq, rp_ in genes["V"]:
        relpos[name] = rp_
        assert rp_ + len(seq) >= fb["v_r"][0]
    for name, seq, rp_ in genes["D"]:
        relpos[name] = rp_
        assert rp_ <= fb["d_l"][1] and rp_ + len(seq) >= fb["d_r"][0]
    for name, seq, rp_ in genes["J"]:
        relpos[name] = rp_
        assert rp_ <= fb["j_l"][1]
    aligned = [to_str(seqs[i]) for i in range(1, T)]
    n_ind = min(int(spec.shm_indels), spec.n_leaves)
    cluster = {
        "germline-info": {"locus": spec.locus},
        "events": [{
            "input_seqs": [a[:L // 2] + a[L // 2 + 1:] if i < n_ind else a for i, a in enumerate(aligned)],
            "indel_reversed_seqs": [a if i < n_ind else "" for i, a in enumerate(aligned)],
            "naive_seq": to_str(naive),
            "has_shm_indels": [i < n_ind for i in range(spec.n_leaves)],
            "linearham-info": {"relpos": relpos, "flexbounds": {k: list(v) for k, v in fb.items()}},
            "unique_ids": labels[1:],
        }],
    }
    with open(os.path.join(outdir, "cluster.yaml"), "w") as f:
        json.dump(cluster, f, indent=1)
    # RevBayes-style tree samples
    srng = np.random.default_rng(spec.seed + 1)
    with open(os.path.join(outdir, "trees.tsv"), "w") as f:
        cols = ["Iteration", "Posterior", "Likelihood", "Prior", "alpha"] + ["er[%d]" % i for i in range(1, 7)] \
            + ["pi[%d]" % i for i in range(1, 5)] + ["tree"]
        f.write("\t".join(cols) + "\n")
        for s in range(spec.n_samples):
            e2 = {k: list(v) for k, v in edges.items()}
            for _ in range(spec.n_nni):
                nni(e2, T, srng)
            for v in e2.values():
                v[2] = max(v[2] * float(srng.lognormal(0.0, 0.3)), 1e-6)
            er = srng.dirichlet(np.ones(6))
            pi = srng.dirichlet(np.ones(4))
            pi = np.clip(pi, 0.01, None)
            pi /= pi.sum()
            alpha = max(float(srng.exponential(1.0)), 0.05)
            lik = -1000.0 - 10.0 * srng.random()
            row = [str(s * 10), "%.4f" % (lik - 50), "%.4f" % lik, "-50.0", "%.8g" % alpha] + \
                ["%.8g" % x for x in er] + ["%.8g" % x for x in pi] + [newick(e2, T, labels)]
            f.write("\t".join(row) + "\n")
    meta = {"n_leaves": spec.n_leaves, "n_sites": L, "n_v": spec.n_v, "n_d": spec.n_d, "n_j": spec.n_j,
            "n_samples": spec.n_samples, "seed": spec.seed, "flexbounds": {k: list(v) for k, v in fb.items()}}
    with open(os.path.join(outdir, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    return meta


def read_trees_tsv(path, max_rows=None):
    """Minimal reader of the RevBayes table (columns by name, extra columns ignored)."""
    rows = []
    with open(path) as f:
        header = f.readline().rstrip("\n").split("\t")
        ix = {h: i for i, h in enumerate(header)}
        for line in f:
            if max_rows is not None and len(rows) >= max_rows:
                break
            p = line.rstrip("\n").split("\t")
            rows.append({"iteration": int(p[ix["Iteration"]]), "likelihood": float(p[ix["Likelihood"]]),
                         "prior": float(p[ix["Prior"]]), "alpha": float(p[ix["alpha"]]),
                         "er": [float(p[ix["er[%d]" % i]]) for i in range(1, 7)],
                         "pi": [float(p[ix["pi[%d]" % i]]) for i in range(1, 5)],
                         "tree": p[ix["tree"]].strip('"')})
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--preset", default="config2", choices=["config2", "config3", "config4", "small"])
    ap.add_argument("--n-samples", type=int, default=None)
    a = ap.parse_args()
    spec = {"config2": Spec(), "config3": Spec(n_samples=10000), "small": Spec.small(),
            "config4": Spec(n_leaves=500, n_sites=600)}[a.preset]
    if a.n_samples:
        spec.n_samples = a.n_samples
    print(json.dumps(generate(spec, a.outdir)))

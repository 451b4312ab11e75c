"""TEST INFRASTRUCTURE: build the C-ABI family descriptor (linearham_amd.capi.FamilyDesc) and the
rooted-at-naive tree arrays from an *oracle* PhyloHMM/SimpleHMM object, plus a numpy emulation of
the kernels' structured algorithm (used on CPU to validate the descriptor semantics against the
dense oracle before any GPU time is spent).  The product builds the same descriptor in C++
(linearham_amd/csrc/host); both are checked against the dense oracle."""
import math

import numpy as np

from linearham_amd.capi import FamilyDesc, JunctionTables, Segments
from oracle import linearham_oracle as orc


def _segments(R, inds):
    offs = [0]
    out = []
    for gname in sorted(R.ggene_ranges):
        rs, re_ = R.ggene_ranges[gname]
        out.extend(int(x) for x in inds[rs:re_])
        offs.append(len(out))
    return Segments(offs, out)


def junction_tables(h, J, Jx, G_left, G_right, left_fb):
    """J: junction Region; Jx: its W x S xMSA index matrix; G_left/G_right: neighbouring germline
    Regions."""
    js = left_fb[0]
    W = Jx.shape[0]
    left = sorted(G_left.ggene_ranges)
    right = sorted(G_right.ggene_ranges)
    nL, nR = len(left), len(right)
    t = dict(enter_trans=np.zeros(nL), enter_lo=np.zeros(nL), left_trans=np.zeros((W, nL)),
             left_lo=np.zeros((W, nL)), left_xmsa=-np.ones((W, nL), dtype=np.int32),
             right_gp_nli=np.zeros((nR, 4)), right_ntt=np.zeros((nR, 4, 4)), right_nlo=np.zeros((W, nR, 4)),
             right_trans=np.zeros((W, nR)), right_gp_li=np.zeros((W, nR)),
             right_xmsa=-np.ones((W, nR), dtype=np.int32), nti_xmsa=-np.ones((W, nR, 4), dtype=np.int32),
             exit_nlo=np.zeros((nR, 4)), exit_trans=np.zeros(nR), exit_gp_li=np.zeros(nR))
    for l, name in enumerate(left):
        gg = h.ggenes[name]
        frs, fre = G_left.ggene_ranges[name]
        p_last = G_left.germ_inds[fre - 1]
        t["enter_lo"][l] = gg.landing_out[p_last]
        rs, re_ = J.ggene_ranges[name]
        cnt = re_ - rs
        if cnt > 0:
            t["enter_trans"][l] = gg.transition[p_last]
        for i in range(cnt):
            p = J.germ_inds[rs + i]
            assert J.site_inds[rs + i] == js + i
            if i >= 1:
                t["left_trans"][i, l] = gg.transition[p - 1]
            t["left_lo"][i, l] = gg.landing_out[p]
            t["left_xmsa"][i, l] = Jx[i, rs + i]
    for r, name in enumerate(right):
        gg = h.ggenes[name]
        rs, re_ = J.ggene_ranges[name]
        t["right_gp_nli"][r] = gg.gene_prob * gg.nti_landing_in
        t["right_ntt"][r] = gg.nti_transition
        for b in range(4):
            for i in range(W):
                t["nti_xmsa"][i, r, b] = Jx[i, rs + b]
        first = True
        last_row = -1
        for k in range(rs + 4, re_):
            q = J.germ_inds[k]
            i = J.site_inds[k] - js
            t["right_nlo"][i, r] = gg.nti_landing_out[:, q]
            if not first:
                t["right_trans"][i, r] = gg.transition[q - 1]
            t["right_gp_li"][i, r] = gg.gene_prob * gg.landing_in[q]
            t["right_xmsa"][i, r] = Jx[i, k]
            first = False
            last_row = i
        trs, tre = G_right.ggene_ranges[name]
        q0 = G_right.germ_inds[trs]
        prod = float(np.prod(gg.transition[q0:q0 + (tre - trs - 1)]))
        t["exit_nlo"][r] = gg.nti_landing_out[:, q0] * prod
        if last_row == W - 1:
            t["exit_trans"][r] = gg.transition[q0 - 1] * prod
        t["exit_gp_li"][r] = gg.gene_prob * gg.landing_in[q0] * prod
    return JunctionTables(W, nL, nR, **t)


def build_family_desc(h):
    """h: oracle PhyloHMM (after construction)."""
    C = h.xmsa.shape[1]
    xmsa_site = np.zeros(C, dtype=np.int32)
    xmsa_base = np.zeros(C, dtype=np.uint8)
    for (base, site), xi in h.xmsa_ids.items():
        xmsa_site[xi] = site
        xmsa_base[xi] = base
    fb = h.flexbounds
    igh = h.locus == "igh"
    vnames = sorted(h.vgerm.ggene_ranges)
    gp = [h.ggenes[g].gene_prob for g in vnames]
    prod = []
    for g in vnames:
        rs, re_ = h.vgerm.ggene_ranges[g]
        gis = h.vgerm.germ_inds[rs]
        prod.append(float(np.prod(h.ggenes[g].transition[gis:gis + (re_ - rs - 1)])))
    if igh:
        vd = junction_tables(h, h.vd_junction, h.vd_junction_xmsa_inds, h.vgerm, h.dgerm, fb["v_r"])
        dj = junction_tables(h, h.dj_junction, h.dj_junction_xmsa_inds, h.dgerm, h.jgerm, fb["d_r"])
        dgerm = _segments(h.dgerm, h.dgerm_xmsa_inds)
    else:
        vd = junction_tables(h, h.vd_junction, h.vd_junction_xmsa_inds, h.vgerm, h.jgerm, fb["v_r"])
        dj = None
        dgerm = None
    return FamilyDesc(
        has_d=igh, msa=h.msa.astype(np.uint8), xmsa_site=xmsa_site, xmsa_naive_base=xmsa_base,
        vpadding=_segments(h.vpadding, h.vpadding_xmsa_inds), vgerm=_segments(h.vgerm, h.vgerm_xmsa_inds),
        dgerm=dgerm, jgerm=_segments(h.jgerm, h.jgerm_xmsa_inds),
        jpadding=_segments(h.jpadding, h.jpadding_xmsa_inds), vgerm_gene_prob=gp,
        vpadding_transition=h.vpadding_transition, vgerm_trans_prod=prod,
        jpadding_transition=h.jpadding_transition, vd=vd, dj=dj)


def tree_arrays(tree, xmsa_labels):
    """oracle Tree -> (children [(T-2)*2], root, brlen [2T-2]) in the C-ABI's rooted-at-naive form:
    tip ids follow xmsa_labels (0 = naive), inner nodes are renumbered T.. in DFS order."""
    T = tree.n_tips
    lab2id = {lab: i for i, lab in enumerate(xmsa_labels)}
    assert sorted(tree.labels) == sorted(xmsa_labels), (tree.labels, xmsa_labels)
    naive = tree.labels.index("naive")
    (root_old, naive_len), = tree.adj[naive]
    newid = {}
    for i, lab in enumerate(tree.labels):
        newid[i] = lab2id[lab]
    nxt = [T]
    children = np.zeros((T - 2, 2), dtype=np.int32)
    brlen = np.zeros(2 * T - 2)
    brlen[0] = naive_len

    def visit(node, par):
        if node >= T:
            newid[node] = nxt[0]
            nxt[0] += 1
            kids = [(nb, l) for nb, l in tree.adj[node] if nb != par]
            assert len(kids) == 2
            for k, (nb, l) in enumerate(kids):
                visit(nb, node)
                children[newid[node] - T, k] = newid[nb]
                brlen[newid[nb]] = l
    import sys
    sys.setrecursionlimit(10000)
    visit(root_old, naive)
    return children.ravel(), newid[root_old], brlen


# ------------------------------------------------------------------------------------------------
# numpy emulation of the kernels' algorithm (CPU validation of the descriptor semantics)
# ------------------------------------------------------------------------------------------------

def emulate_prune(desc, n_tips, ops, brlen, er, pi, rates):
    """K0b + K1 + K2a in numpy: returns em[C]."""
    T = n_tips
    L = desc.msa.shape[1]
    R = len(rates)
    P = orc.gtr_pmatrices(er, pi, rates, brlen)          # [2T-2, R, 4, 4]
    site_lik = np.zeros((R, 5, L))
    onehot = np.concatenate([np.eye(4), np.ones((1, 4))], axis=0)
    for r in range(R):
        def tipvec(tip):
            st = desc.msa[tip - 1]
            return (P[tip, r] @ onehot[st].T)            # [4, L]
        acc = None
        stack = {}
        for op in ops:
            kind, push = op[0] & 15, op[0] & 16
            if push:
                stack[op[3]] = acc
            if kind == 0:
                acc = tipvec(op[1]) * tipvec(op[2])
            elif kind == 1:
                acc = tipvec(op[1]) * (P[op[2], r] @ acc)
            else:
                acc = (P[op[1], r] @ stack[op[3]]) * (P[op[2], r] @ acc)
        w = np.asarray(pi)[:, None] * acc
        for b in range(5):
            site_lik[r, b] = (w * (P[0, r] @ onehot[b])[:, None]).sum(axis=0)
    lik = site_lik.mean(axis=0)                           # equal weights
    em = np.zeros(desc.n_xmsa)
    for c in range(desc.n_xmsa):
        b, s = desc.xmsa_naive_base[c], desc.xmsa_site[c]
        lnl = math.log(lik[b, s])
        if b != 4:
            lnl -= math.log(pi[b])
        em[c] = math.exp(lnl)
    return em


def _fill_segments(seg, em):
    n = len(seg.offsets) - 1
    out, cnt = np.ones(n), [0] * n
    for g in range(n):
        v, c = 1.0, 0
        for j in range(seg.offsets[g], seg.offsets[g + 1]):
            v *= em[seg.xmsa_inds[j]]
            while 0 < v < orc.SCALE_THRESHOLD:
                v *= orc.SCALE_FACTOR
                c += 1
        out[g], cnt[g] = v, c
    mx = max(cnt) if cnt else 0
    for g in range(n):
        d = mx - cnt[g]
        out[g] *= math.pow(orc.SCALE_FACTOR, d) if d < 4 else math.inf
    return out, mx


def _junction(J, em, g_in, count, germ_em, pad_trans, pad_em):
    E = lambda idx: np.where(idx >= 0, em[np.maximum(idx, 0)], 0.0)
    fL = fN = fR = None
    for i in range(J.n_rows):
        if i == 0:
            prevL, lo = g_in, J.enter_lo
            A = float(np.sum(prevL * lo))
            fL2 = prevL * J.enter_trans * E(J.left_xmsa[0])
            fN2 = (A * J.right_gp_nli) * E(J.nti_xmsa[0])
            fR2 = (A * J.right_gp_li[0]) * E(J.right_xmsa[0])
        else:
            A = float(np.sum(fL * J.left_lo[i - 1]))
            fL2 = fL * J.left_trans[i] * E(J.left_xmsa[i])
            fN2 = (np.einsum("rb,rbc->rc", fN, J.right_ntt) + A * J.right_gp_nli) * E(J.nti_xmsa[i])
            fR2 = (np.einsum("rb,rb->r", fN, J.right_nlo[i]) + fR * J.right_trans[i] + A * J.right_gp_li[i]) \
                * E(J.right_xmsa[i])
        allv = np.concatenate([fL2.ravel(), fN2.ravel(), fR2.ravel()])
        k = orc.scale_matrix(allv)
        sc = orc.SCALE_FACTOR ** k if k < 4 else math.inf
        fL, fN, fR = fL2 * sc, fN2 * sc, fR2 * sc
        count += k
    A = float(np.sum(fL * J.left_lo[J.n_rows - 1]))
    g = (np.einsum("rb,rb->r", fN, J.exit_nlo) + fR * J.exit_trans + A * J.exit_gp_li) * germ_em
    if pad_trans is not None:
        g = g * pad_trans
    if pad_em is not None:
        g = g * pad_em
    g = g.copy()
    count += orc.scale_matrix(g)
    return g, count


def emulate_forward(desc, em):
    """K2b in numpy: returns log-likelihood."""
    vp, c1 = _fill_segments(desc.vpadding, em)
    vg, c2 = _fill_segments(desc.vgerm, em)
    f = desc.vgerm_gene_prob * desc.vpadding_transition * vp * desc.vgerm_trans_prod * vg
    f = f.copy()
    vcount = c1 + c2 + orc.scale_matrix(f)
    if desc.has_d:
        de, c = _fill_segments(desc.dgerm, em)
        g, dcount = _junction(desc.vd, em, f, vcount, de, None, None)
        dcount += c
        je, c1 = _fill_segments(desc.jgerm, em)
        jp, c2 = _fill_segments(desc.jpadding, em)
        g, jcount = _junction(desc.dj, em, g, dcount, je, desc.jpadding_transition, jp)
        jcount += c1 + c2
    else:
        je, c1 = _fill_segments(desc.jgerm, em)
        jp, c2 = _fill_segments(desc.jpadding, em)
        g, jcount = _junction(desc.vd, em, f, vcount, je, desc.jpadding_transition, jp)
        jcount += c1 + c2
    return math.log(g.sum()) - jcount * orc.LOG_SCALE_FACTOR

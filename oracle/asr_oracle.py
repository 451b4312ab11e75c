"""CPU restatement (numpy) of linearham's ancestral-sequence step -- TEST INFRASTRUCTURE ONLY.

Follows scripts/run_bootstrap_asr_ess.R:48-104 of matsengrp/linearham: for one tree sample (tree, er,
pi, the R site rates written in the sr[] columns, the sampled NaiveSequence) and every alignment site
  1. the likelihood of the column (naive base on the `naive` tip) on each rate-scaled tree
     (`phylomd::phylo.likelihood(sr.trees[[k]], subst.mod, msa[, j]) / naive.probs[j]`, :79-81),
  2. one rate category drawn with those weights (`sample(..., prob = sr.probs)`, :82),
  3. one joint draw of the states of all inner nodes given the tips on that tree (`phylomd::asr.sim`,
     :84) -- the root state from pi_i * (partial likelihood of the data at i), then every node given its
     parent's state from P(parent -> child)[s_parent][c] * (partial likelihood below the child)(c).

PARITY STATUS: **unpinned**.  phylomd is an R package that is neither in /root/reference nor in this
image, the reference's tests hold no fixture for this step, and R's Mersenne-Twister / `sample()` stream
(run on a `parallel` cluster whose workers share one seed, :44-46) is not reproducible here.  What this file
pins instead: (a) the conditional distributions, checked against brute-force enumeration of the joint
posterior on small trees (tests/test_asr_oracle.py); (b) the site likelihoods per rate, which are the
pruning values already pinned by the reference's goldens (oracle/linearham_oracle.py).  Random numbers are a
counter-based Philox4x32-10 stream keyed by (seed; sample, site, draw) so that GPU and oracle can be
compared draw by draw; known-answer vectors of the generator are checked in the same test file.

Tree form: the C ABI's rooted-at-naive arrays (include/linearham_amd.h, lh_schedule_tree): tips 0..T-1
(0 = naive, i = MSA row i-1), inner nodes T..2T-3, `root` = naive's neighbour, children[2*(v-T)+{0,1}],
brlen[v] = branch above v.  ape::root(tree, "naive", resolve.root = TRUE) (:53) adds a root node on the
naive branch at distance 0 from naive's neighbour, whose state therefore equals that node's state.
"""
import numpy as np

from . import linearham_oracle as orc

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11; Random123).  Inputs: arrays/scalars of 32-bit values.
    Returns four uint64 arrays holding 32-bit words."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & MASK for x in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def uniform(seed, sample, site, draw):
    """53-bit uniform in [0, 1) of the (sample, site, draw) cell of stream `seed`."""
    o0, o1, _, _ = philox4x32_10(site, draw, int(sample) & 0xFFFFFFFF, int(sample) >> 32,
                                 int(seed) & 0xFFFFFFFF, int(seed) >> 32)
    return ((o0 >> np.uint64(5)) * np.uint64(67108864) + (o1 >> np.uint64(6))).astype(np.float64) * (1.0 / 9007199254740992.0)


def draw(weights, u):
    """Index of the first category whose cumulative weight exceeds u * total (weights [..., K], u [...]);
    the last category if none does (only possible when every weight is zero)."""
    w = np.asarray(weights, dtype=np.float64)
    cum = np.cumsum(w, axis=-1)       # left-to-right sums, as the kernel forms them
    t = u * cum[..., -1]
    idx = np.sum(cum <= t[..., None], axis=-1)
    return np.minimum(idx, w.shape[-1] - 1)


def upward(children, root, brlen, T, tip_states, P):
    """Partial likelihoods below every inner node for one rate.  tip_states [T, L] (0..4), P [2T-2, 4, 4].
    Returns clv [2T-2, L, 4] (tips as one-hot / ones; rescaled freely -- only ratios within a node matter)
    and the per-site log of the scale removed (so that site likelihoods can be formed)."""
    L = tip_states.shape[1]
    onehot = np.concatenate([np.eye(4), np.ones((1, 4))], axis=0)
    clv = np.zeros((2 * T - 2, L, 4))
    logscale = np.zeros(L)
    for t in range(T):
        clv[t] = onehot[tip_states[t]]
    order = []
    stack = [root]
    while stack:
        v = stack.pop()
        order.append(v)
        for c in children[2 * (v - T):2 * (v - T) + 2]:
            if c >= T:
                stack.append(c)
    for v in reversed(order):
        a, b = children[2 * (v - T)], children[2 * (v - T) + 1]
        x = np.einsum("ij,lj->li", P[a], clv[a]) * np.einsum("ij,lj->li", P[b], clv[b])
        m = x.max(axis=1)
        m = np.where(m > 0, m, 1.0)
        clv[v] = x / m[:, None]
        logscale += np.log(m)
    return clv, logscale, order


def asr_sample(children, root, brlen, T, msa, naive, er, pi, rates, seed, sample_index):
    """One tree sample.  msa [T-1, L] ints 0..4 (row i = tip i+1), naive [L] ints 0..4.
    Returns (rate_choice [L], anc [T-2, L] states of inner nodes T..2T-3, detail dict)."""
    children = np.asarray(children).ravel()
    msa = np.asarray(msa)
    naive = np.asarray(naive)
    L = msa.shape[1]
    R = len(rates)
    pi = np.asarray(pi, dtype=float)
    tips = np.concatenate([naive[None, :], msa], axis=0)
    P = orc.gtr_pmatrices(er, pi, rates, brlen, small_qt_form=True)          # [2T-2, R, 4, 4]
    sites = np.arange(L)
    per_rate = []
    loglik = np.zeros((R, L))
    for k in range(R):
        clv, logscale, order = upward(children, root, brlen, T, tips, P[:, k])
        # close the naive branch: L = sum_i pi_i clv_root[i] (P_naive clv_naive)[i]
        down = np.einsum("ij,lj->li", P[0, k], clv[0])
        w_root = pi[None, :] * clv[root] * down
        with np.errstate(divide="ignore"):     # (a category under which a column is impossible: weight 0)
            loglik[k] = np.log(w_root.sum(axis=1)) + logscale
        per_rate.append((clv, w_root, order))
    # 1-2: rate category per site (the division by naive.probs is common to the R weights)
    w = np.exp(loglik - loglik.max(axis=0, keepdims=True)).T          # [L, R]
    rate_choice = draw(w, uniform(seed, sample_index, sites, 0))
    # 3: joint draw of the inner states on the chosen tree
    anc = np.zeros((T - 2, L), dtype=np.uint8)
    cond = {}
    for k in range(R):
        sel = np.nonzero(rate_choice == k)[0]
        if sel.size == 0:
            continue
        clv, w_root, order = per_rate[k]
        s_root = draw(w_root[sel], uniform(seed, sample_index, sel, 1))
        anc[root - T, sel] = s_root
        for v in order:                      # pre-order: parents before children
            sp = anc[v - T, sel]
            for c in children[2 * (v - T):2 * (v - T) + 2]:
                if c < T:
                    continue
                wc = P[c, k][sp, :] * clv[c][sel]            # P(parent -> child)[s_p][.] * partial(.)
                anc[c - T, sel] = draw(wc, uniform(seed, sample_index, sel, 2 + (c - T)))
    return rate_choice.astype(np.uint8), anc, {"loglik_per_rate": loglik}


def exact_joint_posterior(children, root, brlen, T, column, er, pi, rate):
    """Brute force over all 4^(T-2) inner-state assignments for ONE column (column[0] = naive's state,
    0..4) on the tree scaled by `rate`: the joint posterior asr.sim samples from.  Small T only."""
    children = np.asarray(children).ravel()
    P = orc.gtr_pmatrices(er, pi, [rate], brlen)[:, 0]
    I = T - 2
    onehot = np.concatenate([np.eye(4), np.ones((1, 4))], axis=0)
    post = np.zeros((4,) * I)
    for idx in np.ndindex(*post.shape):
        st = dict((T + i, s) for i, s in enumerate(idx))
        p = pi[st[root]] * (P[0][st[root]] @ onehot[column[0]])
        for v in range(T, 2 * T - 2):
            for c in children[2 * (v - T):2 * (v - T) + 2]:
                p *= (P[c][st[v]] @ onehot[column[c]]) if c < T else P[c][st[v], st[c]]
        post[idx] = p
    return post / post.sum()

/* ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of the per-tree-sample evaluation of the reference, algorithm for algorithm
 * (dense, as the reference executes it), used (a) as a second, independent checker next to
 * oracle/linearham_oracle.py and (b) as bench.py's `cpu_baseline` ("port": the reference binary
 * cannot be built offline).  Citations are file:line into matsengrp/linearham.
 *
 *   per sample:  P-matrices            pll_update_prob_matrices [3P], src/PhyloHMM.cpp:225
 *                pruning, all C xMSA columns, per-site 2^256 scaling
 *                                      pll_update_partials / edge log-likelihood [3P], :225-226
 *                naive correction+exp  PhyloHMM::FillXmsaEmission, :229-237
 *                emission fills        FillGermlinePaddingEmission :158-193, FillJunctionEmission :202-215
 *                forward               src/HMM.cpp:291-319, 1107-1139, 1160-1177 (dense rowvec x matrix)
 *                log-likelihood        src/HMM.cpp:345-354
 *
 * Parity status: pinned through tests/test_oracle_c.py (agrees with the numpy oracle, which is pinned
 * to the reference's Catch-test literals, to <= 1e-12 relative on the toy and synthetic families).
 * Discrete-Gamma rates are computed by the caller (scipy) and passed in.
 *
 * ext != 0 (oc_eval_batch_ext): the product's opt-in extended-range mode (include/linearham_amd.h,
 * lh_family_set_extended_range) restated on the same dense algorithm -- NOT reference behaviour: emissions
 * are (value, 2^-256 count) pairs, a region's products are equalised to the smallest count, vectors are
 * rescaled by their largest entry.  Parity of that mode is therefore "unpinned" by construction; it is
 * checked against this file where the reference is finite (same log-likelihood) and for finiteness elsewhere.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SCALE_FACTOR 0x1p256
#define SCALE_THRESHOLD 0x1p-256
#define LOG_SCALE_FACTOR 177.445678223345993274

typedef struct {
  int n_genes;
  const int32_t* offsets;   /* [n_genes+1] */
  const int32_t* xmsa_inds; /* concatenated per gene, std::map order */
} oc_segments;

typedef struct {
  int W, S, n_from, n_to;
  const double* T_gj;   /* [n_from][S]  row-major */
  const double* T_jj;   /* [S][S]       column-major (Eigen default): T_jj[c*S + r] */
  const double* T_jg;   /* [S][n_to]    column-major: T_jg[c*S + r] */
  const int32_t* xmsa;  /* [W][S] row-major, -1 = structural zero */
} oc_junction;

typedef struct {
  int T, C, R, has_d;
  const uint8_t* xmsa;            /* [T][C], row 0 = naive; 4 = N */
  oc_segments vpadding, vgerm, dgerm, jgerm, jpadding;
  const double* vgerm_gene_prob;  /* [nV] */
  const double* vpadding_transition;
  const double* vgerm_trans_prod;
  const double* jpadding_transition;
  oc_junction vd, dj;
} oc_family;

static int scale_vec(double* v, int n) { /* ScaleMatrix, src/utils.cpp:135-144 */
  int k = 0;
  for (;;) {
    int any = 0;
    for (int i = 0; i < n; ++i)
      if (v[i] > 0.0 && v[i] < SCALE_THRESHOLD) { any = 1; break; }
    if (!any) return k;
    for (int i = 0; i < n; ++i) v[i] *= SCALE_FACTOR;
    ++k;
  }
}

/* symmetric 4x4 Jacobi eigendecomposition */
static void jacobi4(double A[4][4], double W[4][4]) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) W[i][j] = (i == j);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int i = 0; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        double c = 1 / sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 4; ++k) {
          double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 4; ++k) {
          double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 4; ++k) {
          double wkp = W[k][p], wkq = W[k][q];
          W[k][p] = c * wkp - s * wkq;
          W[k][q] = s * wkp + c * wkq;
        }
      }
  }
}

static int scale_vec_max(double* v, int n) { /* extended-range mode: by the largest entry */
  int k = 0;
  for (;;) {
    double mx = 0;
    for (int i = 0; i < n; ++i)
      if (v[i] > mx) mx = v[i];
    if (!(mx > 0.0 && mx < SCALE_THRESHOLD)) return k;
    for (int i = 0; i < n; ++i) v[i] *= SCALE_FACTOR;
    ++k;
  }
}

static int fill_segments(const oc_segments* s, const double* em, const int* emc, double* out) {
  /* FillGermlinePaddingEmission, src/PhyloHMM.cpp:158-193; emc != NULL: extended-range mode */
  int mx = 0, mn = 1 << 30;
  int* cnt = (int*)malloc(sizeof(int) * (s->n_genes > 0 ? s->n_genes : 1));
  for (int g = 0; g < s->n_genes; ++g) {
    double v = 1.0;
    int c = 0;
    for (int j = s->offsets[g]; j < s->offsets[g + 1]; ++j) {
      v *= em[s->xmsa_inds[j]];
      if (emc) c += emc[s->xmsa_inds[j]];
      c += scale_vec(&v, 1);
    }
    out[g] = v;
    cnt[g] = c;
    if (c > mx) mx = c;
    if (v > 0.0 && c < mn) mn = c;
  }
  if (emc) {
    if (mn == 1 << 30) mn = 0;
    for (int g = 0; g < s->n_genes; ++g)
      for (int d = cnt[g] - mn; d > 0 && out[g] != 0.0; --d) out[g] *= SCALE_THRESHOLD;
    free(cnt);
    return mn;
  }
  for (int g = 0; g < s->n_genes; ++g) out[g] *= pow(SCALE_FACTOR, mx - cnt[g]);
  free(cnt);
  return mx;
}

/* ComputeJunctionForwardProbabilities + ComputeGermlineForwardProbabilities for one junction */
static int junction(const oc_junction* J, const double* em, const int* emc, const double* g_in, int count_in,
                    const double* germ_em, const double* pad_trans, const double* pad_em, double* g_out,
                    double* buf /* 3*S */, double* rows_out /* W*S or NULL */, int* counts_out /* W or NULL */) {
  const int S = J->S, W = J->W;
  double* prev = buf;
  double* cur = buf + S;
  double* E = buf + 2 * S;
  int count = count_in;
  for (int i = 0; i < W; ++i) {
    for (int s = 0; s < S; ++s) { /* FillJunctionEmission row */
      const int idx = J->xmsa[(size_t)i * S + s];
      E[s] = idx >= 0 ? em[idx] : 0.0;
    }
    if (emc) { /* extended range: bring the row's emissions to their smallest count, which joins the row's */
      int mn = 1 << 30;
      for (int s = 0; s < S; ++s) {
        const int idx = J->xmsa[(size_t)i * S + s];
        if (idx >= 0 && em[idx] > 0.0 && emc[idx] < mn) mn = emc[idx];
      }
      if (mn == 1 << 30) mn = 0;
      for (int s = 0; s < S; ++s) {
        const int idx = J->xmsa[(size_t)i * S + s];
        if (idx >= 0)
          for (int d = emc[idx] - mn; d > 0 && E[s] != 0.0; --d) E[s] *= SCALE_THRESHOLD;
      }
      count += mn;
    }
    if (i == 0) {
      for (int s = 0; s < S; ++s) {
        double acc = 0;
        for (int f = 0; f < J->n_from; ++f) acc += g_in[f] * J->T_gj[(size_t)f * S + s];
        cur[s] = acc;
      }
    } else {
      for (int s = 0; s < S; ++s) {
        const double* col = J->T_jj + (size_t)s * S;
        double acc = 0;
        for (int r = 0; r < S; ++r) acc += prev[r] * col[r];
        cur[s] = acc;
      }
    }
    for (int s = 0; s < S; ++s) cur[s] *= E[s];
    count += emc ? scale_vec_max(cur, S) : scale_vec(cur, S);
    if (rows_out) memcpy(rows_out + (size_t)i * S, cur, sizeof(double) * S); /* junction_forward_.row(i), src/HMM.cpp:1133-1137 */
    if (counts_out) counts_out[i] = count;                                   /* junction_scaler_counts_[i] */
    double* t = prev; prev = cur; cur = t;
  }
  for (int g = 0; g < J->n_to; ++g) {
    const double* col = J->T_jg + (size_t)g * S;
    double acc = 0;
    for (int r = 0; r < S; ++r) acc += prev[r] * col[r];
    acc *= germ_em[g];
    if (pad_trans) acc *= pad_trans[g];
    if (pad_em) acc *= pad_em[g];
    g_out[g] = acc;
  }
  return count + (emc ? scale_vec_max(g_out, J->n_to) : scale_vec(g_out, J->n_to));
}

/* One evaluation.  children/root/brlen in the C ABI's rooted-at-naive form (any rooting gives the
 * same likelihood; the naive tip is an ordinary tip of every xMSA column here). order = inner nodes
 * in post-order.  Returns log-likelihood; writes em[C] if non-NULL. */
/* fwd_out / cnt_out (optional): the forward arrays SampleNaiveSequence reads (src/HMM.cpp:326,1250,1333), dense:
 * vgerm[nV] | vd junction [W][S] | dgerm[nD] | dj junction [W][S] | jgerm[nJ] (light chains: vgerm | junction | jgerm)
 * and the scaler counts vgerm | vd rows | dgerm | dj rows | jgerm. */
static double eval_one(const oc_family* F, const int32_t* children, int root, const int32_t* order,
                       const double* brlen, const double* er, const double* pi, const double* rates,
                       double* em_out, int ext, double* fwd_out, int* cnt_out) {
  const int T = F->T, C = F->C, R = F->R, nodes = 2 * T - 2, I = T - 2;
  /* GTR eigendecomposition */
  double S[4][4] = {{0}}, A[4][4], W[4][4], sq[4], lam[4], U[4][4], Ui[4][4];
  S[0][1] = S[1][0] = er[0]; S[0][2] = S[2][0] = er[1]; S[0][3] = S[3][0] = er[2];
  S[1][2] = S[2][1] = er[3]; S[1][3] = S[3][1] = er[4]; S[2][3] = S[3][2] = er[5];
  double mu = 0, diag[4];
  for (int i = 0; i < 4; ++i) {
    double rs = 0;
    for (int j = 0; j < 4; ++j) if (j != i) rs += S[i][j] * pi[j];
    diag[i] = -rs; mu += pi[i] * rs; sq[i] = sqrt(pi[i]);
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) A[i][j] = (i == j) ? diag[i] / mu : S[i][j] * sq[i] * sq[j] / mu;
  jacobi4(A, W);
  for (int k = 0; k < 4; ++k) lam[k] = A[k][k];
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 4; ++k) { U[i][k] = W[i][k] / sq[i]; Ui[k][i] = W[i][k] * sq[i]; }
  /* P-matrices [node][rate][4][4] */
  double* P = (double*)malloc(sizeof(double) * (size_t)nodes * R * 16);
  for (int v = 0; v < nodes; ++v)
    for (int r = 0; r < R; ++r) {
      double ex[4];
      /* libpll's form for Qt -> 0 (core_pmatrix.c, [3P]): expm1 of the eigenvalues, the identity added at the end */
      for (int k = 0; k < 4; ++k) ex[k] = expm1(lam[k] * brlen[v] * rates[r]);
      double* p = P + ((size_t)v * R + r) * 16;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          double acc = (i == j) ? 1.0 : 0.0;
          for (int k = 0; k < 4; ++k) acc += U[i][k] * ex[k] * Ui[k][j];
          p[i * 4 + j] = acc;
        }
    }
  /* CLVs [inner][rate][4][C] with per-site scalers; tips handled on the fly */
  double* clv = (double*)malloc(sizeof(double) * (size_t)I * R * 4 * C);
  int32_t* scal = (int32_t*)calloc((size_t)I * C, sizeof(int32_t));
  double tmp[2][4];
  for (int oi = 0; oi < I; ++oi) {
    const int v = order[oi];
    double* out = clv + (size_t)(v - T) * R * 4 * C;
    int32_t* sc = scal + (size_t)(v - T) * C;
    for (int c = 0; c < C; ++c) {
      double mx = 0;
      for (int r = 0; r < R; ++r) {
        for (int side = 0; side < 2; ++side) {
          const int ch = children[2 * (v - T) + side];
          const double* p = P + ((size_t)ch * R + r) * 16;
          if (ch < T) {
            const int st = F->xmsa[(size_t)ch * C + c];
            for (int i = 0; i < 4; ++i)
              tmp[side][i] = st < 4 ? p[i * 4 + st] : (p[i * 4] + p[i * 4 + 1] + p[i * 4 + 2] + p[i * 4 + 3]);
          } else {
            const double* cc = clv + ((size_t)(ch - T) * R + r) * 4 * C;
            for (int i = 0; i < 4; ++i)
              tmp[side][i] = p[i * 4] * cc[c] + p[i * 4 + 1] * cc[C + c] + p[i * 4 + 2] * cc[2 * C + c] +
                             p[i * 4 + 3] * cc[3 * C + c];
          }
        }
        for (int i = 0; i < 4; ++i) {
          const double x = tmp[0][i] * tmp[1][i];
          out[((size_t)r * 4 + i) * C + c] = x;
          if (x > mx) mx = x;
        }
      }
      int s = 0;
      for (int side = 0; side < 2; ++side) {
        const int ch = children[2 * (v - T) + side];
        if (ch >= T) s += scal[(size_t)(ch - T) * C + c];
      }
      if (mx < SCALE_THRESHOLD && mx > 0) {
        for (int r = 0; r < R; ++r)
          for (int i = 0; i < 4; ++i) out[((size_t)r * 4 + i) * C + c] *= SCALE_FACTOR;
        ++s;
      }
      sc[c] = s;
    }
  }
  /* root edge (root -- naive tip 0), naive correction, exp */
  double* em = (double*)malloc(sizeof(double) * C);
  int* emc = ext ? (int*)calloc(C, sizeof(int)) : NULL;
  {
    const double* rc = clv + (size_t)(root - T) * R * 4 * C;
    for (int c = 0; c < C; ++c) {
      const int st = F->xmsa[c]; /* row 0 = naive */
      double site = 0;
      for (int r = 0; r < R; ++r) {
        const double* p = P + ((size_t)0 * R + r) * 16;
        double acc = 0;
        for (int i = 0; i < 4; ++i) {
          const double tp = st < 4 ? p[i * 4 + st] : (p[i * 4] + p[i * 4 + 1] + p[i * 4 + 2] + p[i * 4 + 3]);
          acc += pi[i] * rc[((size_t)r * 4 + i) * C + c] * tp;
        }
        site += acc / R;
      }
      if (ext) { /* value and 2^-256 count side by side, no exp */
        em[c] = st != 4 ? site / pi[st] : site;
        emc[c] = scal[(size_t)(root - T) * C + c];
        continue;
      }
      double lnl = log(site) - scal[(size_t)(root - T) * C + c] * LOG_SCALE_FACTOR;
      if (st != 4) lnl -= log(pi[st]);
      em[c] = exp(lnl);
    }
  }
  free(P); free(clv); free(scal);
  if (em_out) memcpy(em_out, em, sizeof(double) * C);
  /* emissions + forward */
  int mg = F->vgerm.n_genes;
  if (F->dgerm.n_genes > mg) mg = F->dgerm.n_genes;
  if (F->jgerm.n_genes > mg) mg = F->jgerm.n_genes;
  int Smax = F->vd.S > F->dj.S ? F->vd.S : F->dj.S;
  double* e1 = (double*)malloc(sizeof(double) * mg * 4);
  double *e2 = e1 + mg, *gA = e1 + 2 * mg, *gB = e1 + 3 * mg;
  double* buf = (double*)malloc(sizeof(double) * 3 * (size_t)Smax);
  const int nV = F->vgerm.n_genes;
  int vcount = fill_segments(&F->vpadding, em, emc, e2);
  vcount += fill_segments(&F->vgerm, em, emc, e1);
  for (int g = 0; g < nV; ++g) { /* ComputeInitialForwardProbabilities, src/HMM.cpp:291-319 */
    double v = F->vgerm_gene_prob[g];
    v *= F->vpadding_transition[g];
    v *= e2[g];
    v *= F->vgerm_trans_prod[g];
    v *= e1[g];
    gA[g] = v;
  }
  vcount += ext ? scale_vec_max(gA, nV) : scale_vec(gA, nV);
  double* fo = fwd_out;
  int* co = cnt_out;
  if (fo) { memcpy(fo, gA, sizeof(double) * nV); fo += nV; }
  if (co) *co++ = vcount;
  int jcount;
  const double* gJ;
  if (F->has_d) {
    int dcount = fill_segments(&F->dgerm, em, emc, e1);
    dcount += junction(&F->vd, em, emc, gA, vcount, e1, NULL, NULL, gB, buf, fo, co);
    if (fo) { fo += (size_t)F->vd.W * F->vd.S; memcpy(fo, gB, sizeof(double) * F->dgerm.n_genes); fo += F->dgerm.n_genes; }
    if (co) { co += F->vd.W; *co++ = dcount; }
    jcount = fill_segments(&F->jgerm, em, emc, e1);
    jcount += fill_segments(&F->jpadding, em, emc, e2);
    jcount += junction(&F->dj, em, emc, gB, dcount, e1, F->jpadding_transition, e2, gA, buf, fo, co);
    if (fo) fo += (size_t)F->dj.W * F->dj.S;
    if (co) co += F->dj.W;
    gJ = gA;
  } else {
    jcount = fill_segments(&F->jgerm, em, emc, e1);
    jcount += fill_segments(&F->jpadding, em, emc, e2);
    jcount += junction(&F->vd, em, emc, gA, vcount, e1, F->jpadding_transition, e2, gB, buf, fo, co);
    if (fo) fo += (size_t)F->vd.W * F->vd.S;
    if (co) co += F->vd.W;
    gJ = gB;
  }
  if (fo) memcpy(fo, gJ, sizeof(double) * F->jgerm.n_genes);
  if (co) *co = jcount;
  double tot = 0;
  for (int g = 0; g < F->jgerm.n_genes; ++g) tot += gJ[g];
  const double ll = log(tot) - jcount * LOG_SCALE_FACTOR;
  free(em); free(emc); free(e1); free(buf);
  return ll;
}

/* n samples; arrays laid out as for lh_eval_batch plus order[n][T-2] (post-order of inner nodes) and
 * rates[n][R].  n_threads > 1 uses OpenMP over samples. */
int oc_eval_batch(const oc_family* F, int n, const int32_t* children, const int32_t* roots,
                  const int32_t* order, const double* brlen, const double* er, const double* pi,
                  const double* rates, double* loglik, double* em_out, int n_threads) {
  const int T = F->T;
#pragma omp parallel for schedule(dynamic) num_threads(n_threads > 0 ? n_threads : 1)
  for (int s = 0; s < n; ++s)
    loglik[s] = eval_one(F, children + (size_t)s * 2 * (T - 2), roots[s], order + (size_t)s * (T - 2),
                         brlen + (size_t)s * (2 * T - 2), er + (size_t)s * 6, pi + (size_t)s * 4,
                         rates + (size_t)s * F->R, em_out ? em_out + (size_t)s * F->C : NULL, 0, NULL, NULL);
  return 0;
}

/* Sizes of the dense forward output of oc_eval_batch_fwd (doubles / ints per sample). */
int64_t oc_forward_size(const oc_family* F) {
  int64_t n = F->vgerm.n_genes + (int64_t)F->vd.W * F->vd.S + F->jgerm.n_genes;
  if (F->has_d) n += F->dgerm.n_genes + (int64_t)F->dj.W * F->dj.S;
  return n;
}
int64_t oc_counts_size(const oc_family* F) { return 2 + F->vd.W + (F->has_d ? 1 + F->dj.W : 0); }

/* oc_eval_batch that also returns what sampling reads: forward[n][oc_forward_size] and counts[n][oc_counts_size]
 * (layout at eval_one). */
int oc_eval_batch_fwd(const oc_family* F, int n, const int32_t* children, const int32_t* roots,
                      const int32_t* order, const double* brlen, const double* er, const double* pi,
                      const double* rates, double* loglik, double* forward, int32_t* counts, int n_threads) {
  const int T = F->T;
  const int64_t fs = oc_forward_size(F), cs = oc_counts_size(F);
#pragma omp parallel for schedule(dynamic) num_threads(n_threads > 0 ? n_threads : 1)
  for (int s = 0; s < n; ++s)
    loglik[s] = eval_one(F, children + (size_t)s * 2 * (T - 2), roots[s], order + (size_t)s * (T - 2),
                         brlen + (size_t)s * (2 * T - 2), er + (size_t)s * 6, pi + (size_t)s * 4,
                         rates + (size_t)s * F->R, NULL, 0, forward + (size_t)s * fs, (int*)counts + (size_t)s * cs);
  return 0;
}

/* The same in the product's extended-range mode (see the header comment). */
int oc_eval_batch_ext(const oc_family* F, int n, const int32_t* children, const int32_t* roots,
                      const int32_t* order, const double* brlen, const double* er, const double* pi,
                      const double* rates, double* loglik, int n_threads) {
  const int T = F->T;
#pragma omp parallel for schedule(dynamic) num_threads(n_threads > 0 ? n_threads : 1)
  for (int s = 0; s < n; ++s)
    loglik[s] = eval_one(F, children + (size_t)s * 2 * (T - 2), roots[s], order + (size_t)s * (T - 2),
                         brlen + (size_t)s * (2 * T - 2), er + (size_t)s * 6, pi + (size_t)s * 4,
                         rates + (size_t)s * F->R, NULL, 1, NULL, NULL);
  return 0;
}

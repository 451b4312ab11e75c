// K0: substitution-model set-up kernels (gfx950).
//
// Replaces, per tree sample (citations into matsengrp/linearham; [3P] = libpll via libptpll):
//   pll_compute_gamma_cats(alpha, R, rates, PLL_GAMMA_RATES_MEAN)        src/PhyloHMM.cpp:425-426 [3P]
//   pt::pll::Model{"GTR", pi, er, sr} + eigendecomposition in Partition   src/PhyloHMM.cpp:368-370 [3P]
//
// Tiny next to the pruning kernel (36 + R doubles per sample); the P-matrices themselves are computed
// by K1's prologue (compute_pmatrix, lh_device.h).
#include "lh_device.h"

namespace lh {

// ---- regularised incomplete gamma and its inverse (double) ---------------------------------------

// (fast_rcp, lh_device.h: the series and the continued fraction below are chains of dependent divisions, and the
// IEEE division sequence is four times as long.)

__device__ static double gamma_p(double a, double x, double lga) {
  if (!(x > 0.0)) return 0.0;
  const double pre = exp(-x + a * log(x) - lga);
  if (x < a + 1.0) {  // series
    double ap = a, del = fast_rcp(a), sum = del;
    for (int n = 0; n < 2000; ++n) {
      ap += 1.0;
      del *= x * fast_rcp(ap);
      sum += del;
      if (fabs(del) < fabs(sum) * 1e-17) break;
    }
    return sum * pre;
  }
  // continued fraction for Q (modified Lentz)
  const double FPMIN = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / FPMIN, d = fast_rcp(b), h = d;
  for (int i = 1; i < 2000; ++i) {
    const double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < FPMIN) d = FPMIN;
    c = b + an * fast_rcp(c);
    if (fabs(c) < FPMIN) c = FPMIN;
    d = fast_rcp(d);
    const double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return 1.0 - pre * h;
}

// x such that P(a, x) = p  (Halley iteration from the Numerical-Recipes starting guess)
__device__ static double gamma_p_inv(double p, double a) {
  const double gln = lgamma(a);
  const double a1 = a - 1.0;
  double lna1 = 0.0, afac = 0.0, x, t;
  if (a > 1.0) {
    lna1 = log(a1);
    afac = exp(a1 * (lna1 - 1.0) - gln);
    const double pp = (p < 0.5) ? p : 1.0 - p;
    t = sqrt(-2.0 * log(pp));
    x = (2.30753 + t * 0.27061) / (1.0 + t * (0.99229 + t * 0.04481)) - t;
    if (p < 0.5) x = -x;
    const double w = 1.0 - 1.0 / (9.0 * a) - x / (3.0 * sqrt(a));
    x = fmax(1e-3, a * w * w * w);
  } else {
    t = 1.0 - a * (0.253 + a * 0.12);
    if (p < t)
      x = pow(p / t, 1.0 / a);
    else
      x = 1.0 - log(1.0 - (p - t) / (1.0 - t));
  }
  for (int j = 0; j < 32; ++j) {
    if (!(x > 1e-300)) return 0.0;
    const double err = gamma_p(a, x, gln) - p;
    if (a > 1.0)
      t = afac * exp(-(x - a1) + a1 * (log(x) - lna1));
    else
      t = exp(-x + a1 * log(x) - gln);
    const double u = err / t;
    t = u / (1.0 - 0.5 * fmin(1.0, u * (a1 / x - 1.0)));
    x -= t;
    if (x <= 0.0) x = 0.5 * (x + t);
    if (fabs(t) < 1e-11 * x) break;  // Halley converges cubically: the next step would move x by < 1e-30 x
  }
  return x;
}

// ---- K0a -------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(64) model_setup_kernel(int n, int R, const double* __restrict__ er,
                                                         const double* __restrict__ pi,
                                                         const double* __restrict__ alpha,
                                                         double* __restrict__ rates,
                                                         double* __restrict__ eig) {
  // R threads per sample, role-major so that a wave is homogeneous: roles 0..R-2 each solve ONE
  // category boundary of the discrete Gamma, role R-1 does the GTR eigendecomposition.
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)n * R) return;
  const int role = (int)(gid / n);
  const int s = (int)(gid % n);

  // (a3) discrete-Gamma category means, equal weights: r_k = R [P(a+1, y_k) - P(a+1, y_{k-1})],
  // y_k = a * (k/R quantile of Gamma(a, rate a)).  Here only the cumulative P(a+1, y_k) is produced
  // (into rates[s][k-1]); finalize_rates_kernel turns the cumulatives into category means.
  if (role < R - 1) {
    const double a = alpha[s];
    const double y = gamma_p_inv((double)(role + 1) / R, a);
    rates[(size_t)s * R + role] = gamma_p(a + 1.0, y, lgamma(a + 1.0));
    return;
  }

  // (a4) GTR: Q_ij = er_ij pi_j (AC,AG,AT,CG,CT,GT), diagonal = -row sum, scaled to mean rate 1.
  // Symmetrise A = Pi^{1/2} Q Pi^{-1/2}, cyclic Jacobi, U = Pi^{-1/2} W, Uinv = W^T Pi^{1/2}.
  double p[4], sq[4];
  for (int i = 0; i < 4; ++i) {
    p[i] = pi[(size_t)s * 4 + i];
    sq[i] = sqrt(p[i]);
  }
  double S[4][4];
  {
    const double* e = er + (size_t)s * 6;
    S[0][1] = S[1][0] = e[0];
    S[0][2] = S[2][0] = e[1];
    S[0][3] = S[3][0] = e[2];
    S[1][2] = S[2][1] = e[3];
    S[1][3] = S[3][1] = e[4];
    S[2][3] = S[3][2] = e[5];
    S[0][0] = S[1][1] = S[2][2] = S[3][3] = 0.0;
  }
  double diag[4], mu = 0.0;
  for (int i = 0; i < 4; ++i) {
    double rs = 0.0;
    for (int j = 0; j < 4; ++j)
      if (j != i) rs += S[i][j] * p[j];
    diag[i] = -rs;
    mu += p[i] * rs;
  }
  double A[4][4], W[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      A[i][j] = (i == j) ? diag[i] / mu : S[i][j] * sq[i] * sq[j] / mu;
      W[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
    if (off < 1e-300) break;
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
#pragma unroll
      for (int q = pp + 1; q < 4; ++q) {
        const double apq = A[pp][q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[q][q] - A[pp][pp]) / (2.0 * apq);
        const double t = ((theta >= 0.0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // A <- A J
          const double akp = A[k][pp], akq = A[k][q];
          A[k][pp] = c * akp - sn * akq;
          A[k][q] = sn * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // A <- J^T A
          const double apk = A[pp][k], aqk = A[q][k];
          A[pp][k] = c * apk - sn * aqk;
          A[q][k] = sn * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // W <- W J
          const double wkp = W[k][pp], wkq = W[k][q];
          W[k][pp] = c * wkp - sn * wkq;
          W[k][q] = sn * wkp + c * wkq;
        }
      }
    }
  }
  // The stationary mode (eigenvalue 0: the largest, every other one is negative) goes first: its term of
  // P = I + U expm1(lambda t) U^-1 is expm1(0) = 0, and compute_pmatrix (lh_device.h) leaves it out.
  // (Bubbled down with statically indexed conditional swaps: a run-time column index would put W in scratch.)
  double lam[4] = {A[0][0], A[1][1], A[2][2], A[3][3]};
#pragma unroll
  for (int k = 3; k >= 1; --k) {
    bool up = true;  // is lam[k] the largest of lam[0..k]?
#pragma unroll
    for (int j = 0; j < k; ++j) up = up && lam[k] > lam[j];
    if (up) {
      const double t = lam[k];
      lam[k] = lam[k - 1];
      lam[k - 1] = t;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double w = W[i][k];
        W[i][k] = W[i][k - 1];
        W[i][k - 1] = w;
      }
    }
  }
  double* o = eig + (size_t)s * 36;
  o[0] = 0.0;
  for (int k = 1; k < 4; ++k) o[k] = lam[k];
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 4; ++k) {
      o[4 + i * 4 + k] = W[i][k] / sq[i];   // U[i][k]
      o[20 + k * 4 + i] = W[i][k] * sq[i];  // Uinv[k][i]
    }
}

__global__ void __launch_bounds__(256) finalize_rates_kernel(int n, int R, double* __restrict__ rates) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  double* r = rates + (size_t)s * R;
  double prev = 0.0;
  for (int k = 0; k < R - 1; ++k) {
    const double cum = r[k];
    r[k] = (cum - prev) * R;
    prev = cum;
  }
  r[R - 1] = (1.0 - prev) * R;   // R == 1: the single rate is 1
}

void launch_model_setup(int n, int R, const double* er, const double* pi, const double* alpha,
                        double* rates, double* eig, hipStream_t stream) {
  const long long total = (long long)n * R;
  hipLaunchKernelGGL(model_setup_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, stream, n, R, er, pi,
                     alpha, rates, eig);
  hipLaunchKernelGGL(finalize_rates_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, R, rates);
}

void launch_gtr_setup(int n, const double* er, const double* pi, double* eig, hipStream_t stream) {
  // R = 1: every thread takes the eigendecomposition role, no Gamma boundary is solved
  hipLaunchKernelGGL(model_setup_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, n, 1, er, pi,
                     (const double*)nullptr, (double*)nullptr, eig);
}

}  // namespace lh

// pca.hip -- streaming PCA fit: one pass of covariance accumulation on the f64 matrix unit.
// Replaces the per-iteration `IncrementalPCA.partial_fit` loop of the reference (_dimensionality.py:69-108, 171-241: a
// forked process per iteration, an SVD of (components + batch) x features per call) by sums the whole chip forms while the
// features stream through HBM; the D x D eigen-decomposition that turns the sums into components stays on the host
// (msm_we_amd/pca.py), like every solve of this package.  Opt-in (`model.pca_fit = "device"`): the default remains the
// reference's own estimator on the host, whose truncated updates are not a function of the covariance alone.
//
// Arithmetic contract (include/hamsm.h; restated by oracle/hamsm_oracle.c: bit-identical):
//   shift   mu0 = frame 0 of the first non-empty hamsm_pca_accumulate after hamsm_pca_begin
//   z(f)    = ((double)x_a(f) - mu0_a for a < D,  1 for a = D)            -- the ones column carries S1 and n
//   call    the n frames are cut into G strips of L frames (pca_strips below); per strip and pair a <= b the products
//           z_a(f) z_b(f) are accumulated in frame order as ONE fma chain from 0 -- exactly what v_mfma_f64_16x16x4_f64 does
//           with frames as its k dimension (hamsm_selftest_mfma) --, the strips' sums are added in strip order inside groups
//           of 64 consecutive strips, the groups' sums in group order, and that total is added to the running S2.
#include "common.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

// strips of one accumulate call: enough waves to fill the chip at small D, partial sums within 256 MB at large D
static void pca_strips(int64_t n, int Dp, int *G, int64_t *L) {
  int64_t g = (n + 511) / 512;
  const int64_t cap = ((int64_t)1 << 25) / ((int64_t)Dp * Dp);
  if (g > 8192) g = 8192;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  int64_t l = (n + g - 1) / g;
  l = (l + 3) & ~(int64_t)3;
  *G = (int)g;
  *L = l;
}

// one wave per (32 x 32 block pair bi <= bj, strip): 2 x 2 MFMA tiles, frames 4 at a time as the k dimension, 4 k-steps of
// operands in flight.  Lane l supplies feature (l & 15) of frame (l >> 4) of the quad to both operands.
template <typename XT>
__global__ __launch_bounds__(256) void k_pca_syrk(const XT *__restrict__ X, int64_t ldx, int64_t n, int D, int Dp,
                                                  const double *__restrict__ mu0, int G, int64_t L, int npairs,
                                                  double *__restrict__ P) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
  const long long w = (long long)blockIdx.x * 4 + wv;
  if (w >= (long long)npairs * G) return;
  const int g = (int)(w % G);
  int p = (int)(w / G), bi = 0;
  const int nb = Dp >> 5;
  while (p >= nb - bi) {  // pairs (bi, bj >= bi) in row-major order
    p -= nb - bi;
    bi++;
  }
  const int bj = bi + p;
  const bool diag = bi == bj;
  const int64_t f0 = (int64_t)g * L, fend = f0 + L < n ? f0 + L : n;
  v4f64 acc[2][2];
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int t = 0; t < 2; t++) acc[s][t] = (v4f64){0.0, 0.0, 0.0, 0.0};
  if (f0 < fend) {
    int fa[2], fb[2];       // this lane's features of the two operands
    double ma[2], mb[2];    // their shifts
#pragma unroll
    for (int s = 0; s < 2; s++) {
      fa[s] = bi * 32 + s * 16 + r;
      fb[s] = bj * 32 + s * 16 + r;
      ma[s] = fa[s] < D ? mu0[fa[s]] : 0.0;
      mb[s] = fb[s] < D ? mu0[fb[s]] : 0.0;
    }
    constexpr int U = 4;
    for (int64_t f = f0; f < fend; f += 4 * U) {
      XT xa[U][2], xb[U][2];
#pragma unroll
      for (int i = 0; i < U; i++) {  // unconditional loads at clamped positions; frames / features past the end are deselected below
        const int64_t fr = f + 4 * i + kq, frc = fr < fend ? fr : fend - 1;
        const XT *row = X + frc * ldx;
#pragma unroll
        for (int s = 0; s < 2; s++) {
          xa[i][s] = row[fa[s] < D ? fa[s] : D - 1];
          xb[i][s] = diag ? xa[i][s] : row[fb[s] < D ? fb[s] : D - 1];  // (wave-uniform: a diagonal block loads its rows once)
        }
      }
#pragma unroll
      for (int i = 0; i < U; i++) {
        const bool in = f + 4 * i + kq < fend;
        double za[2], zb[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const double va = fa[s] < D ? (double)xa[i][s] - ma[s] : (fa[s] == D ? 1.0 : 0.0);
          const double vb = fb[s] < D ? (double)xb[i][s] - mb[s] : (fb[s] == D ? 1.0 : 0.0);
          za[s] = in ? va : 0.0;  // fma(0, 0, acc) == acc
          zb[s] = in ? vb : 0.0;
        }
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(za[0], zb[0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(za[0], zb[1], acc[0][1], 0, 0, 0);
        if (!diag) acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(za[1], zb[0], acc[1][0], 0, 0, 0);  // (diagonal block: the mirror of [0][1])
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(za[1], zb[1], acc[1][1], 0, 0, 0);
      }
    }
  }
  // lane l holds rows (l >> 4) + 4 q, column l & 15 of each 16 x 16 tile
  double *Pg = P + (size_t)g * Dp * Dp;
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int t = 0; t < 2; t++) {
      if (diag && s == 1 && t == 0) continue;
#pragma unroll
      for (int q = 0; q < 4; q++) Pg[(size_t)(bi * 32 + s * 16 + kq + 4 * q) * Dp + bj * 32 + t * 16 + r] = acc[s][t][q];
    }
}

// the strips' sums are added in strip order inside groups of PCA_GROUP consecutive strips (one thread per entry and group),
// the groups' sums in group order, and that total goes to the running sum: S2[a][b] += (Q_0 + Q_1) + ...,  a <= b
#define PCA_GROUP 64
__global__ void k_pca_reduce1(const double *__restrict__ P, int G, int Dp, double *__restrict__ Q) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, DD = (int64_t)Dp * Dp;
  if (e >= DD) return;
  if ((int)(e / Dp) > (int)(e % Dp)) return;
  const int g0 = blockIdx.y * PCA_GROUP, g1 = g0 + PCA_GROUP < G ? g0 + PCA_GROUP : G;
  double t = P[(size_t)g0 * DD + e];
#pragma unroll 8
  for (int g = g0 + 1; g < g1; g++) t = t + P[(size_t)g * DD + e];
  Q[(size_t)blockIdx.y * DD + e] = t;
}
__global__ void k_pca_reduce2(const double *__restrict__ Q, int NQ, int Dp, double *__restrict__ S2) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, DD = (int64_t)Dp * Dp;
  if (e >= DD) return;
  if ((int)(e / Dp) > (int)(e % Dp)) return;
  double t = Q[e];
#pragma unroll 8
  for (int q = 1; q < NQ; q++) t = t + Q[(size_t)q * DD + e];
  S2[e] = S2[e] + t;
}

template <typename XT>
__global__ void k_pca_shift(const XT *__restrict__ X, int D, double *__restrict__ mu0) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < D) mu0[a] = (double)X[a];
}

static void pca_free(hamsm_ctx *c) {
  if (c->pca_S2) hipFree(c->pca_S2);
  if (c->pca_mu0) hipFree(c->pca_mu0);
  c->pca_S2 = c->pca_mu0 = nullptr;
  c->pca_D = 0;
  c->pca_n = 0;
}

extern "C" int hamsm_pca_begin(hamsm_ctx *c, int Draw) {
  CTX(c);
  if (Draw < 1 || Draw > HAMSM_MAX_D) return hamsm_fail(c, HAMSM_EINVAL, "pca_begin: %d features", Draw);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  pca_free(c);
  const int Dp = (Draw + 1 + 31) & ~31;
  HIPCHK(c, hipMalloc((void **)&c->pca_S2, sizeof(double) * (size_t)Dp * Dp));
  HIPCHK(c, hipMalloc((void **)&c->pca_mu0, sizeof(double) * Dp));
  HIPCHK(c, hipMemsetAsync(c->pca_S2, 0, sizeof(double) * (size_t)Dp * Dp, c->stream));
  HIPCHK(c, hipMemsetAsync(c->pca_mu0, 0, sizeof(double) * Dp, c->stream));
  c->pca_D = Draw;
  return HAMSM_OK;
}

extern "C" int hamsm_pca_accumulate(hamsm_ctx *c, const void *X, int dt, int64_t ldx, int64_t n) {
  CTX(c);
  if (!c->pca_S2) return hamsm_fail(c, HAMSM_EINVAL, "pca_accumulate before pca_begin");
  if (dt != HAMSM_F32 && dt != HAMSM_F64) return hamsm_fail(c, HAMSM_EINVAL, "bad x_dtype");
  const int D = c->pca_D, Dp = (D + 1 + 31) & ~31;
  if (ldx < D) return hamsm_fail(c, HAMSM_EINVAL, "ldx < D");
  if (n <= 0) return HAMSM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int G;
  int64_t L;
  pca_strips(n, Dp, &G, &L);
  const int NQ = (G + PCA_GROUP - 1) / PCA_GROUP;
  int rc = hamsm_ensure_scratch(c, sizeof(double) * (size_t)(G + NQ) * Dp * Dp);
  if (rc) return rc;
  double *P = (double *)c->scratch, *Q = P + (size_t)G * Dp * Dp;
  const int nb = Dp >> 5, npairs = nb * (nb + 1) / 2;
  const long long waves = (long long)npairs * G;
  hamsm_timer_begin(c);
  if (c->pca_n == 0) {
    if (dt == HAMSM_F32) hipLaunchKernelGGL(k_pca_shift<float>, dim3((D + 255) / 256), dim3(256), 0, c->stream, (const float *)X, D, c->pca_mu0);
    else hipLaunchKernelGGL(k_pca_shift<double>, dim3((D + 255) / 256), dim3(256), 0, c->stream, (const double *)X, D, c->pca_mu0);
  }
  // (off-diagonal tiles below the diagonal blocks are never written nor read: the reductions visit a <= b only)
  const dim3 grid((unsigned)((waves + 3) / 4));
  if (dt == HAMSM_F32)
    hipLaunchKernelGGL(k_pca_syrk<float>, grid, dim3(256), 0, c->stream, (const float *)X, ldx, n, D, Dp, c->pca_mu0, G, L, npairs, P);
  else
    hipLaunchKernelGGL(k_pca_syrk<double>, grid, dim3(256), 0, c->stream, (const double *)X, ldx, n, D, Dp, c->pca_mu0, G, L, npairs, P);
  const unsigned ge = (unsigned)(((size_t)Dp * Dp + 255) / 256);
  hipLaunchKernelGGL(k_pca_reduce1, dim3(ge, (unsigned)NQ), dim3(256), 0, c->stream, P, G, Dp, Q);
  hipLaunchKernelGGL(k_pca_reduce2, dim3(ge), dim3(256), 0, c->stream, Q, NQ, Dp, c->pca_S2);
  hamsm_timer_end(c, ST_OTHER);
  HIPCHK(c, hipGetLastError());
  c->pca_n += n;
  return HAMSM_OK;
}

extern "C" int hamsm_pca_finish(hamsm_ctx *c, int64_t *n_total, double *h_shift, double *h_S1, double *h_S2) {
  CTX(c);
  if (!c->pca_S2) return hamsm_fail(c, HAMSM_EINVAL, "pca_finish before pca_begin");
  const int D = c->pca_D, Dp = (D + 1 + 31) & ~31;
  std::vector<double> S((size_t)Dp * Dp), mu(Dp);
  HIPCHK(c, hipMemcpyAsync(S.data(), c->pca_S2, sizeof(double) * S.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(mu.data(), c->pca_mu0, sizeof(double) * Dp, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n_total) *n_total = c->pca_n;
  for (int a = 0; a < D; a++) {
    if (h_shift) h_shift[a] = mu[a];
    if (h_S1) h_S1[a] = S[(size_t)a * Dp + D];  // the ones column
    if (h_S2)
      for (int b = 0; b < D; b++) h_S2[(size_t)a * D + b] = a <= b ? S[(size_t)a * Dp + b] : S[(size_t)b * Dp + a];
  }
  pca_free(c);
  return HAMSM_OK;
}

void hamsm_pca_release(hamsm_ctx *c) { pca_free(c); }

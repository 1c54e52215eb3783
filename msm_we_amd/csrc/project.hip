// project.hip -- a2: the linear map of the dimensionality reduction, y = x W^T - mean W^T (/ scale): what sklearn's
// PCA / IncrementalPCA .transform computes (sklearn/decomposition/_base.py `_transform`), called per iteration by the
// reference (reduceCoordinates, _dimensionality.py:37-66; _clustering.py:894-896, 1291-1296).  Here: once per frame at
// staging time, f64 MFMA with the components as the A operand (packed like the centroid operands of assign.hip) and 16
// frames per wave as the B operand; each output is an in-order fma chain over the raw features, as in the oracle.
#include "common.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

// Wp[jt][ks][lane]: component jt*16 + (lane & 15), raw feature 4 ks + (lane >> 4)
__global__ void k_pack_w(const double *__restrict__ W, int d, int Draw, int KS, int njt, double *__restrict__ Wp) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)njt * KS * 64) return;
  int lane = (int)(e & 63);
  int ks = (int)((e >> 6) % KS), jt = (int)((e >> 6) / KS);
  int j = jt * 16 + (lane & 15), k = 4 * ks + (lane >> 4);
  Wp[e] = (j < d && k < Draw) ? W[(size_t)j * Draw + k] : 0.0;
}

// One workgroup = 128 frames (4 waves x 2 tiles of 16).  The raw features stream through LDS in chunks of 32 (8 k-steps):
// coalesced global loads (a wave reads two rows' 128-byte pieces), row stride 34 elements so that both the staging
// writes and the fragment reads (16 frames x 2 features per half-wave) hit 32 distinct banks; the component fragments
// of the chunk are staged once per workgroup instead of being fetched from L2 by every wave and k-step.  Chunk c+1 is
// requested before chunk c is consumed.  Up to four component tiles per sweep over the features.
#define PJ_FR 128
#define PJ_KB 32
#define PJ_XS 34
template <typename XT>
__global__ __launch_bounds__(256) void k_project(const XT *__restrict__ X, int64_t ldx, int64_t n, int Draw, int KS,
                                                 const double *__restrict__ Wp, int njt, const double *__restrict__ m,
                                                 const double *__restrict__ scale, int d, double *__restrict__ Y) {
  __shared__ XT xs[2][PJ_FR * PJ_XS];
  __shared__ double ws[2][4 * 8 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t f0 = (int64_t)blockIdx.x * PJ_FR;
  const int nch = (Draw + PJ_KB - 1) / PJ_KB;
  for (int j0 = 0; j0 < njt; j0 += 4) {
    const int nj = njt - j0 < 4 ? njt - j0 : 4;
    v4f64 acc[4][2];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int t = 0; t < 2; t++) acc[u][t] = (v4f64){0.0, 0.0, 0.0, 0.0};
    XT xr[PJ_FR * PJ_KB / 256];
    double wr[4 * 8 * 64 / 256];
    auto fetch = [&](int c) {  // chunk c: features [32 c, 32 c + 32) of the 128 rows, k-steps [8 c, 8 c + 8) of nj tiles
#pragma unroll
      for (int i = 0; i < PJ_FR * PJ_KB / 256; i++) {
        const int e = tid + 256 * i, r = e / PJ_KB, k = c * PJ_KB + e % PJ_KB;
        const int64_t f = f0 + r;
        xr[i] = (f < n && k < Draw) ? X[f * ldx + k] : (XT)0;
      }
#pragma unroll
      for (int i = 0; i < 4 * 8 * 64 / 256; i++) {
        const int e = tid + 256 * i, u = e / 512, ks = c * 8 + (e % 512) / 64;
        wr[i] = (u < nj && ks < KS) ? Wp[((size_t)(j0 + u) * KS + ks) * 64 + (e & 63)] : 0.0;
      }
    };
    auto stash = [&](int buf) {
#pragma unroll
      for (int i = 0; i < PJ_FR * PJ_KB / 256; i++) {
        const int e = tid + 256 * i;
        xs[buf][(e / PJ_KB) * PJ_XS + e % PJ_KB] = xr[i];
      }
#pragma unroll
      for (int i = 0; i < 4 * 8 * 64 / 256; i++) ws[buf][tid + 256 * i] = wr[i];
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (int c = 0; c < nch; c++) {
      const int buf = c & 1;
      if (c + 1 < nch) fetch(c + 1);
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        double b[2];
#pragma unroll
        for (int t = 0; t < 2; t++) b[t] = (double)xs[buf][(wv * 32 + t * 16 + (lane & 15)) * PJ_XS + 4 * ks + (lane >> 4)];
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (u < nj) {
            const double a = ws[buf][(u * 8 + ks) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 2; t++) acc[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[t], acc[u][t], 0, 0, 0);
          }
      }
      if (c + 1 < nch) stash(buf ^ 1);
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int64_t f = f0 + wv * 32 + t * 16 + (lane & 15);
      if (f < n) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int j = (j0 + u) * 16 + (lane >> 4) + 4 * q;
            if (u < nj && j < d) {
              double v = acc[u][t][q];
              if (m) v = v - m[j];          // X_transformed -= mean_ @ components_.T
              if (scale) v = v / scale[j];  // whiten
              Y[(size_t)f * d + j] = v;
            }
          }
      }
    }
    __syncthreads();  // (the next sweep re-stages buffer 0)
  }
}

extern "C" int hamsm_project(hamsm_ctx *c, const void *X, int dt, int64_t ldx, int64_t n, int Draw, const double *h_mean,
                             const double *h_W, int d, const double *h_scale, double *Y) {
  if (Draw < 1 || d < 1 || !h_W || ldx < Draw) return hamsm_fail(c, HAMSM_EINVAL, "bad projection shape (%d x %d, ldx %lld)", d, Draw, (long long)ldx);
  if (dt != HAMSM_F32 && dt != HAMSM_F64) return hamsm_fail(c, HAMSM_EINVAL, "bad x_dtype");
  if (n <= 0) return HAMSM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const int KS = (Draw + 3) / 4, njt = (d + 15) / 16;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t o_w = take(sizeof(double) * (size_t)d * Draw), o_wp = take(sizeof(double) * (size_t)njt * KS * 64),
               o_mean = take(sizeof(double) * Draw), o_m = take(sizeof(double) * d), o_s = take(sizeof(double) * d);
  int rc = hamsm_ensure_scratch2(c, off);
  if (rc) return rc;
  char *b = (char *)c->scratch2;
  HIPCHK(c, hipMemcpyAsync(b + o_w, h_W, sizeof(double) * (size_t)d * Draw, hipMemcpyHostToDevice, c->stream));
  if (h_mean) HIPCHK(c, hipMemcpyAsync(b + o_mean, h_mean, sizeof(double) * Draw, hipMemcpyHostToDevice, c->stream));
  if (h_scale) HIPCHK(c, hipMemcpyAsync(b + o_s, h_scale, sizeof(double) * d, hipMemcpyHostToDevice, c->stream));
  const int64_t ne = (int64_t)njt * KS * 64;
  hamsm_timer_begin(c);
  hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, c->stream, (const double *)(b + o_w), d, Draw, KS,
                     njt, (double *)(b + o_wp));
  const double *Wp = (const double *)(b + o_wp);
  const double *m = nullptr;
  if (h_mean) {  // mean_ @ components_.T by the same chains
    hipLaunchKernelGGL(k_project<double>, dim3(1), dim3(256), 0, c->stream, (const double *)(b + o_mean), (int64_t)Draw, (int64_t)1,
                       Draw, KS, Wp, njt, (const double *)nullptr, (const double *)nullptr, d, (double *)(b + o_m));
    m = (const double *)(b + o_m);
  }
  const double *sc = h_scale ? (const double *)(b + o_s) : nullptr;
  const unsigned grid = (unsigned)((n + PJ_FR - 1) / PJ_FR);
  if (dt == HAMSM_F32)
    hipLaunchKernelGGL(k_project<float>, dim3(grid), dim3(256), 0, c->stream, (const float *)X, ldx, n, Draw, KS, Wp, njt, m, sc, d, Y);
  else
    hipLaunchKernelGGL(k_project<double>, dim3(grid), dim3(256), 0, c->stream, (const double *)X, ldx, n, Draw, KS, Wp, njt, m, sc, d, Y);
  hamsm_timer_end(c, ST_OTHER);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // h_W / h_mean / h_scale are the caller's
  return HAMSM_OK;
}

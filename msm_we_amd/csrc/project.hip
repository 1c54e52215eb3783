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

template <typename XT>
__global__ __launch_bounds__(256) void k_project(const XT *__restrict__ X, int64_t ldx, int64_t n, int Draw, int KS,
                                                 const double *__restrict__ Wp, int njt, const double *__restrict__ m,
                                                 const double *__restrict__ scale, int d, double *__restrict__ Y) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t f = ((int64_t)blockIdx.x * 4 + wv) * 16 + (lane & 15);
  const bool ok = f < n;
  const XT *x = X + (ok ? f : 0) * ldx;
  for (int j0 = 0; j0 < njt; j0 += 4) {  // four component tiles per sweep over the features
    v4f64 acc[4];
#pragma unroll
    for (int u = 0; u < 4; u++) acc[u] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int ks = 0; ks < KS; ks++) {
      const int k = 4 * ks + (lane >> 4);
      const double b = (ok && k < Draw) ? (double)x[k] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (j0 + u < njt) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(Wp[((size_t)(j0 + u) * KS + ks) * 64 + lane], b, acc[u], 0, 0, 0);
    }
    if (ok) {
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int j = (j0 + u) * 16 + (lane >> 4) + 4 * q;
          if (j < d) {
            double v = acc[u][q];
            if (m) v = v - m[j];          // X_transformed -= mean_ @ components_.T
            if (scale) v = v / scale[j];  // whiten
            Y[(size_t)f * d + j] = v;
          }
        }
    }
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
  const unsigned grid = (unsigned)((n + 63) / 64);
  if (dt == HAMSM_F32)
    hipLaunchKernelGGL(k_project<float>, dim3(grid), dim3(256), 0, c->stream, (const float *)X, ldx, n, Draw, KS, Wp, njt, m, sc, d, Y);
  else
    hipLaunchKernelGGL(k_project<double>, dim3(grid), dim3(256), 0, c->stream, (const double *)X, ldx, n, Draw, KS, Wp, njt, m, sc, d, Y);
  hamsm_timer_end(c, ST_OTHER);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // h_W / h_mean / h_scale are the caller's
  return HAMSM_OK;
}

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

// One workgroup = 128 frames (4 waves x 2 tiles of 16).  The raw features stream through LDS in chunks of KB = 32 (64
// was tried: one workgroup per CU, slower): coalesced global loads (thread t always reads column t % KB of rows t / KB + 256/KB i
// -- one base pointer, constant stride), row stride KB + 2 elements so that the staging writes and the fragment reads
// (16 frames x 2 features per half-wave) hit 32 distinct banks; the component fragments of a chunk are staged once per
// workgroup.  Chunk c+1 is requested before chunk c is consumed.  NJ component tiles per sweep over the features is a
// template parameter and every load is unconditional at a clamped address, so the k-step loop has no branches:
// columns >= Draw meet zero fragments (the rows are finite), rows >= n are computed and not stored.
#define PJ_FR 128
#define PJ_KB_OF(XT) 32
template <typename XT, int NJ>
__global__ __launch_bounds__(256) void k_project(const XT *__restrict__ X, int64_t ldx, int64_t n, int Draw, int KS,
                                                 const double *__restrict__ Wp, int njt, const double *__restrict__ m,
                                                 const double *__restrict__ scale, int d, double *__restrict__ Y) {
  constexpr int KB = PJ_KB_OF(XT), XS = KB + 2, KSC = KB / 4, RP = 256 / KB;  // RP rows per staging pass
  constexpr int NX = PJ_FR / RP, NWR = NJ * KSC * 64 / 256;
  // LDS: ws[2][NJ][KSC][64] doubles, then xs[2][128][XS] XT
  extern __shared__ double pj_lds[];
  double *ws0 = pj_lds, *ws1 = pj_lds + (size_t)NJ * KSC * 64;
  XT *xs0 = (XT *)(pj_lds + (size_t)NJ * KSC * 128), *xs1 = xs0 + PJ_FR * XS;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t f0 = (int64_t)blockIdx.x * PJ_FR;
  const int nch = (Draw + KB - 1) / KB;
  const int xr_row = tid / KB, xr_col = tid % KB;
  const bool full = f0 + PJ_FR <= n;
  for (int j0 = 0; j0 < njt; j0 += NJ) {
    v4f64 acc[NJ][2];
#pragma unroll
    for (int u = 0; u < NJ; u++)
#pragma unroll
      for (int t = 0; t < 2; t++) acc[u][t] = (v4f64){0.0, 0.0, 0.0, 0.0};
    XT xr[NX];
    double wr[NWR];
    auto fetch = [&](int c) {
      const int kc = c * KB + xr_col;
      const XT *p = X + (f0 + xr_row) * ldx + (kc < Draw ? kc : Draw - 1);
      if (full) {
#pragma unroll
        for (int i = 0; i < NX; i++) xr[i] = p[(int64_t)(RP * i) * ldx];
      } else {
#pragma unroll
        for (int i = 0; i < NX; i++) {
          const int64_t f = f0 + xr_row + RP * i;
          xr[i] = p[((f < n ? f : n - 1) - (f0 + xr_row)) * ldx];
        }
      }
#pragma unroll
      for (int i = 0; i < NWR; i++) {
        const int e = tid + 256 * i, jt = j0 + e / (KSC * 64), ks = c * KSC + (e % (KSC * 64)) / 64;
        const double v = Wp[((size_t)(jt < njt ? jt : njt - 1) * KS + (ks < KS ? ks : KS - 1)) * 64 + lane];
        wr[i] = (jt < njt && ks < KS) ? v : 0.0;
      }
    };
    auto stash = [&](int buf) {
      XT *xs = buf ? xs1 : xs0;
      double *ws = buf ? ws1 : ws0;
#pragma unroll
      for (int i = 0; i < NX; i++) xs[(xr_row + RP * i) * XS + xr_col] = xr[i];
#pragma unroll
      for (int i = 0; i < NWR; i++) ws[tid + 256 * i] = wr[i];
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (int c = 0; c < nch; c++) {
      const int buf = c & 1;
      const XT *xs = (buf ? xs1 : xs0) + (wv * 32 + (lane & 15)) * XS + (lane >> 4);
      const double *ws = (buf ? ws1 : ws0) + lane;
      if (c + 1 < nch) fetch(c + 1);
#pragma unroll 2
      for (int ks = 0; ks < KSC; ks++) {
        const double b0 = (double)xs[4 * ks], b1 = (double)xs[16 * XS + 4 * ks];
#pragma unroll
        for (int u = 0; u < NJ; u++) {
          const double a = ws[(u * KSC + ks) * 64];
          acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[u][0], 0, 0, 0);
          acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[u][1], 0, 0, 0);
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
        for (int u = 0; u < NJ; u++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int j = (j0 + u) * 16 + (lane >> 4) + 4 * q;
            if (j < d) {
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

template <typename XT>
static int launch_project(hamsm_ctx *c, const XT *X, int64_t ldx, int64_t n, int Draw, int KS, const double *Wp, int njt,
                          const double *m, const double *sc, int d, double *Y) {
  const int nj = njt < 4 ? njt : 4;
  const size_t lds = sizeof(double) * nj * (PJ_KB_OF(XT) / 4) * 128 + sizeof(XT) * 2 * PJ_FR * (PJ_KB_OF(XT) + 2);
  const void *fn = nj == 1 ? (const void *)k_project<XT, 1> : nj == 2 ? (const void *)k_project<XT, 2>
                 : nj == 3 ? (const void *)k_project<XT, 3> : (const void *)k_project<XT, 4>;
  HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  void *args[] = {(void *)&X, (void *)&ldx, (void *)&n, (void *)&Draw, (void *)&KS, (void *)&Wp, (void *)&njt, (void *)&m,
                  (void *)&sc, (void *)&d, (void *)&Y};
  HIPCHK(c, hipLaunchKernel(fn, dim3((unsigned)((n + PJ_FR - 1) / PJ_FR)), dim3(256), args, lds, c->stream));
  return HAMSM_OK;
}

extern "C" int hamsm_project(hamsm_ctx *c, const void *X, int dt, int64_t ldx, int64_t n, int Draw, const double *h_mean,
                             const double *h_W, int d, const double *h_scale, double *Y) {
  CTX(c);
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
    rc = launch_project<double>(c, (const double *)(b + o_mean), (int64_t)Draw, (int64_t)1, Draw, KS, Wp, njt, nullptr, nullptr, d,
                                (double *)(b + o_m));
    if (rc) return rc;
    m = (const double *)(b + o_m);
  }
  const double *sc = h_scale ? (const double *)(b + o_s) : nullptr;
  rc = dt == HAMSM_F32 ? launch_project<float>(c, (const float *)X, ldx, n, Draw, KS, Wp, njt, m, sc, d, Y)
                       : launch_project<double>(c, (const double *)X, ldx, n, Draw, KS, Wp, njt, m, sc, d, Y);
  if (rc) return rc;
  hamsm_timer_end(c, ST_OTHER);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // h_W / h_mean / h_scale are the caller's
  return HAMSM_OK;
}

// ---- a1: distance-to-reference featurizer (include/hamsm.h) -- HBM-bound: a frame's coordinates in, n_sel doubles out.
// blockDim = (ax atoms, 256 / ax frames), ax = the power of two >= n_sel up to 64: a frame's selected atoms are read by
// consecutive lanes (ascending indices: the touched lines of a frame are fetched once), the feature row is written coalesced.
template <typename CT>
__global__ __launch_bounds__(256) void k_featurize_dist(const CT *__restrict__ coords, int64_t ldc, int64_t n,
                                                        const int32_t *__restrict__ idx, int n_sel,
                                                        const float *__restrict__ ref, double *__restrict__ X, int64_t ldx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.y + threadIdx.y;
  if (i >= n) return;
  const CT *fr = coords + i * ldc;
  for (int a = threadIdx.x; a < n_sel; a += blockDim.x) {
    const int at = idx[a];
    const float dx = (float)fr[3 * (int64_t)at] - ref[3 * a], dy = (float)fr[3 * (int64_t)at + 1] - ref[3 * a + 1],
                dz = (float)fr[3 * (int64_t)at + 2] - ref[3 * a + 2];
    double rsq = (double)dx * (double)dx;
    rsq = rsq + (double)dy * (double)dy;
    rsq = rsq + (double)dz * (double)dz;
    X[i * ldx + a] = __builtin_sqrt(rsq);  // (IEEE: correctly rounded, like libm's)
  }
}

extern "C" int hamsm_featurize_dist(hamsm_ctx *c, const void *coords, int dt, int64_t ldc, int64_t n, int n_atoms,
                                    const int32_t *h_idx, int n_sel, const float *h_ref, double *X, int64_t ldx) {
  CTX(c);
  if (dt != HAMSM_F32 && dt != HAMSM_F64) return hamsm_fail(c, HAMSM_EINVAL, "bad c_dtype");
  if (n_atoms < 1 || n_sel < 1 || !h_idx || !h_ref || ldc < 3 * (int64_t)n_atoms || ldx < n_sel)
    return hamsm_fail(c, HAMSM_EINVAL, "bad featurizer shape");
  for (int a = 0; a < n_sel; a++)
    if (h_idx[a] < 0 || h_idx[a] >= n_atoms) return hamsm_fail(c, HAMSM_EINVAL, "atom index %d out of range", (int)h_idx[a]);
  if (n <= 0) return HAMSM_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t o_ref = (sizeof(int32_t) * (size_t)n_sel + 255) & ~(size_t)255;
  int rc = hamsm_ensure_scratch2(c, o_ref + sizeof(float) * 3 * (size_t)n_sel);
  if (rc) return rc;
  char *b = (char *)c->scratch2;
  HIPCHK(c, hipMemcpyAsync(b, h_idx, sizeof(int32_t) * (size_t)n_sel, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_ref, h_ref, sizeof(float) * 3 * (size_t)n_sel, hipMemcpyHostToDevice, c->stream));
  hamsm_timer_begin(c);
  int ax = 1;
  while (ax < n_sel && ax < 64) ax <<= 1;
  const dim3 block(ax, 256 / ax), grid((unsigned)((n + block.y - 1) / block.y));
  if (dt == HAMSM_F32)
    hipLaunchKernelGGL(k_featurize_dist<float>, grid, block, 0, c->stream, (const float *)coords, ldc, n, (const int32_t *)b, n_sel,
                       (const float *)(b + o_ref), X, ldx);
  else
    hipLaunchKernelGGL(k_featurize_dist<double>, grid, block, 0, c->stream, (const double *)coords, ldc, n, (const int32_t *)b, n_sel,
                       (const float *)(b + o_ref), X, ldx);
  hamsm_timer_end(c, ST_OTHER);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // h_idx / h_ref are the caller's
  return HAMSM_OK;
}

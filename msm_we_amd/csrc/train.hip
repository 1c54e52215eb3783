// train.hip -- a5/a6/a7: stratified mini-batch k-means training on device-resident frames.
//   hamsm_train_prepass / hamsm_train_build_rows : planning support for do_stratified_clustering
//                                                   (reference _clustering.py:843-899, quirks Q1/Q2)
//   hamsm_kmeans_seed                             : k-means++ (sklearn _kmeans.py:214-276)
//   hamsm_minibatch_step                          : _labels_inertia + _minibatch_update_dense
//                                                   (_kmeans.py:1620-1642, _k_means_minibatch.pyx:59-108)
//   hamsm_reassign                                : low-count reassignment (_kmeans.py:1652-1673)
// Every reduction follows the order fixed in oracle/hamsm_oracle.c so that results are bit-identical.
#include "common.h"

int hamsm_assign_rows(hamsm_ctx *c, int s, const void *X, int dt, int64_t ldx, const int32_t *rows, int64_t n_b,
                      int32_t *labels);

// ---- prepass: raw bin or -1 (basis/target) per frame + per-iteration histogram --------------------
__global__ void k_prepass(MapperDev m, const double *__restrict__ p, const int64_t *__restrict__ offs, int64_t n_iters,
                          int S, int32_t *__restrict__ bin, unsigned long long *__restrict__ hist, int *err) {
  extern __shared__ unsigned int lh[];  // [S]
  for (int64_t t = blockIdx.x; t < n_iters; t += gridDim.x) {
    for (int s = threadIdx.x; s < S; s += blockDim.x) lh[s] = 0;
    __syncthreads();
    for (int64_t i = offs[t] + threadIdx.x; i < offs[t + 1]; i += blockDim.x) {
      double pi[HAMSM_MAX_NDIM];
      for (int d = 0; d < m.ndim; d++) pi[d] = p[i * m.ndim + d];
      int b;
      if (dev_in_box(m.target, m.ndim, pi) || dev_in_box(m.basis, m.ndim, pi)) b = -1;  // _clustering.py:872-874
      else {
        b = dev_bin_of(m, pi);  // :877 (no we_remap)
        if (b < 0) dev_set_error(err, HAMSM_ERANGE, (int)i, b);
        else atomicAdd(&lh[b], 1u);
      }
      bin[i] = b;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += blockDim.x) hist[t * S + s] = lh[s];
    __syncthreads();
  }
}

extern "C" int hamsm_train_prepass(hamsm_ctx *c, const double *p, int64_t n, const int64_t *h_offs, int64_t n_iters,
                                   int32_t *bin, int64_t *h_hist) {
  (void)n;
  if (!c->have_mapper || !c->have_bounds) return hamsm_fail(c, HAMSM_EINVAL, "mapper/bounds not set");
  if (n_iters <= 0) return HAMSM_OK;
  size_t ob = sizeof(int64_t) * (n_iters + 1), hb = sizeof(unsigned long long) * (size_t)n_iters * c->S;
  size_t o_h = (ob + 255) & ~(size_t)255;
  int rc = hamsm_ensure_scratch(c, o_h + hb);
  if (rc) return rc;
  int64_t *d_offs = (int64_t *)c->scratch;
  unsigned long long *d_hist = (unsigned long long *)((char *)c->scratch + o_h);
  HIPCHK(c, hipMemcpyAsync(d_offs, h_offs, ob, hipMemcpyHostToDevice, c->stream));
  hamsm_timer_begin(c);
  unsigned grid = (unsigned)(n_iters < 4096 ? n_iters : 4096);
  hipLaunchKernelGGL(k_prepass, dim3(grid), dim3(256), sizeof(unsigned int) * c->S, c->stream, c->mapper, p, d_offs, n_iters,
                     c->S, bin, d_hist, c->d_err);
  hamsm_timer_end(c, ST_PREPASS);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(h_hist, d_hist, hb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return hamsm_check_device_error(c, "train_prepass");
}

// ---- row lists of one training call (single workgroup, stable) -------------------------------------
// rank = position in the basis/target-filtered list (:874), applied later to the unfiltered rows (Q1).
#define BR_T 1024
__device__ __forceinline__ int block_excl_scan(int v, int *sh /*[BR_T]*/, int *total) {
  int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int off = 1; off < BR_T; off <<= 1) {
    int x = (tid >= off) ? sh[tid - off] : 0;
    __syncthreads();
    sh[tid] += x;
    __syncthreads();
  }
  int incl = sh[tid];
  *total = sh[BR_T - 1];
  __syncthreads();
  return incl - v;
}

__global__ __launch_bounds__(BR_T) void k_build_rows(const int32_t *__restrict__ bin, int64_t n, const int32_t *__restrict__ cr,
                                                     int S, int32_t *__restrict__ rank, int32_t *__restrict__ rows,
                                                     long long *__restrict__ counts, long long *__restrict__ starts) {
  __shared__ int sh[BR_T];
  const int tid = threadIdx.x;
  const int64_t per = (n + BR_T - 1) / BR_T, a = tid * per, b = (a + per < n) ? a + per : n;
  // ranks among valid frames
  int v = 0;
  for (int64_t i = a; i < b; i++) v += bin[i] >= 0;
  int tot;
  int base = block_excl_scan(v, sh, &tot);
  for (int64_t i = a; i < b; i++)
    if (bin[i] >= 0) rank[i] = base++;
  __syncthreads();
  long long start = 0;
  for (int s = 0; s < S; s++) {
    int cnt = 0;
    for (int64_t i = a; i < b; i++) cnt += (bin[i] >= 0 && cr[bin[i]] == s);
    int ts;
    int o = block_excl_scan(cnt, sh, &ts);
    if (ts) {
      for (int64_t i = a; i < b; i++)
        if (bin[i] >= 0 && cr[bin[i]] == s) rows[start + o++] = rank[i];
    }
    if (tid == 0) {
      counts[s] = ts;
      starts[s] = start;
    }
    start += ts;
  }
}

extern "C" int hamsm_train_build_rows(hamsm_ctx *c, const int32_t *bin, int64_t n, const int32_t *h_cr, int32_t *rows,
                                      int64_t *h_counts, int64_t *h_starts) {
  if (n <= 0) {
    for (int s = 0; s < c->S; s++) h_counts[s] = h_starts[s] = 0;
    return HAMSM_OK;
  }
  size_t o_cr = 0, o_cnt = (sizeof(int32_t) * c->S + 255) & ~(size_t)255, o_st = o_cnt + sizeof(long long) * c->S,
         o_rank = (o_st + sizeof(long long) * c->S + 255) & ~(size_t)255;
  int rc = hamsm_ensure_scratch(c, o_rank + sizeof(int32_t) * (size_t)n);
  if (rc) return rc;
  char *base = (char *)c->scratch;
  HIPCHK(c, hipMemcpyAsync(base + o_cr, h_cr, sizeof(int32_t) * c->S, hipMemcpyHostToDevice, c->stream));
  hamsm_timer_begin(c);
  hipLaunchKernelGGL(k_build_rows, dim3(1), dim3(BR_T), 0, c->stream, bin, n, (const int32_t *)(base + o_cr), c->S,
                     (int32_t *)(base + o_rank), rows, (long long *)(base + o_cnt), (long long *)(base + o_st));
  hamsm_timer_end(c, ST_PREPASS);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(h_counts, base + o_cnt, sizeof(int64_t) * c->S, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_starts, base + o_st, sizeof(int64_t) * c->S, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HAMSM_OK;
}

// ---- k-means++ ------------------------------------------------------------------------------------
template <typename XT>
__global__ void k_seed_gather(const XT *__restrict__ X, int64_t ldx, int D, const int32_t *__restrict__ rows,
                              const long long *__restrict__ sub, int64_t n, const double *__restrict__ wf, double *Xb,
                              double *w, double *xsq) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t r = sub ? sub[i] : i;
  int64_t f = rows ? (int64_t)rows[r] : r;
  double a = 0.0;
  for (int k = 0; k < D; k++) {
    double v = (double)X[f * ldx + k];
    Xb[i * D + k] = v;
    a = __builtin_fma(v, v, a);  // row_norms(X, squared=True) restated as an fma chain (oracle sqnorm_chain)
  }
  xsq[i] = a;
  w[i] = wf ? wf[f] : 1.0;
}

#define SEED_T 256
#define SEED_LMAX 16
// one workgroup runs all K-1 greedy steps; every reduction order is the oracle's
__global__ __launch_bounds__(SEED_T) void k_seed(const double *__restrict__ Xb, const double *__restrict__ w,
                                                 const double *__restrict__ xsq, int64_t n, int D, int K, long long first,
                                                 const double *__restrict__ u, int L, double *closest, double *cand,
                                                 double *cum, double *chunk_tot, double *C, long long *chosen) {
  extern __shared__ double sm[];
  double *crow = sm;                       // [L][D] candidate rows
  double *lpart = crow + SEED_LMAX * D;    // [L][SEED_T] strided partial potentials
  double *pots = lpart + SEED_LMAX * SEED_T;  // [L]
  __shared__ long long cid[SEED_LMAX];
  __shared__ double s_pot;
  __shared__ int s_best;
  const int tid = threadIdx.x;

  // first centre (_kmeans.py:225-238)
  for (int k = tid; k < D; k += SEED_T) {
    crow[k] = Xb[first * D + k];
    C[k] = crow[k];
  }
  if (tid == 0) chosen[0] = first;
  __syncthreads();
  {
    const double xs0 = xsq[first];
    double a = 0.0;
    for (int64_t i = tid; i < n; i += SEED_T) {
      double acc = 0.0;
      for (int k = 0; k < D; k++) acc = __builtin_fma(crow[k], Xb[i * D + k], acc);
      double v = ((-2.0 * acc) + xs0) + xsq[i];
      v = v > 0.0 ? v : 0.0;
      closest[i] = v;
      a = a + v * w[i];
    }
    lpart[tid] = a;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int t = 0; t < SEED_T; t++) s = s + lpart[t];
      s_pot = s;
    }
    __syncthreads();
  }
  const int64_t nch = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  for (int cc = 1; cc < K; cc++) {
    // chunked cumulative sum of w*closest (:243-247 stable_cumsum)
    for (int64_t q = tid; q < nch; q += SEED_T) {
      double loc = 0.0;
      int64_t e = (q + 1) * SCAN_CHUNK < n ? (q + 1) * SCAN_CHUNK : n;
      for (int64_t i = q * SCAN_CHUNK; i < e; i++) loc = loc + w[i] * closest[i];
      chunk_tot[q] = loc;
    }
    __syncthreads();
    if (tid == 0) {
      double base = 0.0;
      for (int64_t q = 0; q < nch; q++) {
        double t = chunk_tot[q];
        chunk_tot[q] = base;  // becomes the chunk's base
        base = base + t;
      }
    }
    __syncthreads();
    for (int64_t q = tid; q < nch; q += SEED_T) {
      double loc = 0.0, base = chunk_tot[q];
      int64_t e = (q + 1) * SCAN_CHUNK < n ? (q + 1) * SCAN_CHUNK : n;
      for (int64_t i = q * SCAN_CHUNK; i < e; i++) {
        loc = loc + w[i] * closest[i];
        cum[i] = base + loc;
      }
    }
    __syncthreads();
    if (tid < L) {
      double v = u[(size_t)(cc - 1) * L + tid] * s_pot;  // :243
      int64_t lo = 0, hi = n;                            // searchsorted side='left'
      while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (cum[mid] < v) lo = mid + 1;
        else hi = mid;
      }
      if (lo > n - 1) lo = n - 1;  // :249
      cid[tid] = lo;
    }
    __syncthreads();
    for (int e = tid; e < L * D; e += SEED_T) crow[e] = Xb[cid[e / D] * D + (e % D)];
    __syncthreads();
    {
      double a[SEED_LMAX];
      for (int l = 0; l < SEED_LMAX; l++) a[l] = 0.0;
      for (int64_t i = tid; i < n; i += SEED_T) {
        double acc[SEED_LMAX];
        for (int l = 0; l < SEED_LMAX; l++) acc[l] = 0.0;
        for (int k = 0; k < D; k++) {
          double xi = Xb[i * D + k];
          for (int l = 0; l < SEED_LMAX; l++)
            if (l < L) acc[l] = __builtin_fma(crow[l * D + k], xi, acc[l]);
        }
        const double ci = closest[i], xi2 = xsq[i], wi = w[i];
        for (int l = 0; l < SEED_LMAX; l++)
          if (l < L) {
            double v = ((-2.0 * acc[l]) + xsq[cid[l]]) + xi2;  // euclidean_distances order (:252-254)
            v = v > 0.0 ? v : 0.0;
            v = v < ci ? v : ci;  // :257 np.minimum
            cand[(size_t)l * n + i] = v;
            a[l] = a[l] + v * wi;
          }
      }
      for (int l = 0; l < L; l++) lpart[l * SEED_T + tid] = a[l];
    }
    __syncthreads();
    if (tid < L) {
      double s = 0.0;
      for (int t = 0; t < SEED_T; t++) s = s + lpart[tid * SEED_T + t];
      pots[tid] = s;  // :258
    }
    __syncthreads();
    if (tid == 0) {
      int best = 0;
      double bp = pots[0];
      for (int l = 1; l < L; l++)
        if (pots[l] < bp) {  // :261 argmin, first minimum
          bp = pots[l];
          best = l;
        }
      s_best = best;
      s_pot = bp;
      chosen[cc] = cid[best];
    }
    __syncthreads();
    const int best = s_best;
    for (int64_t i = tid; i < n; i += SEED_T) closest[i] = cand[(size_t)best * n + i];
    for (int k = tid; k < D; k += SEED_T) C[(size_t)cc * D + k] = crow[best * D + k];
    __syncthreads();
  }
}

extern "C" int hamsm_kmeans_seed(hamsm_ctx *c, int s, const void *X, int dt, int64_t ldx, const int32_t *rows, int64_t n_b,
                                 const double *wf, const int64_t *h_sub, int64_t n_sub, int K, int64_t first,
                                 const double *h_u, int L, int64_t *h_chosen) {
  if (c->st.empty() || s < 0 || s >= c->S) return hamsm_fail(c, HAMSM_EINVAL, "bad stratum");
  const int D = c->D;
  const int64_t n = h_sub ? n_sub : n_b;
  if (K < 1 || n < 1 || first < 0 || first >= n || L < 1 || L > SEED_LMAX) return hamsm_fail(c, HAMSM_EINVAL, "bad seed args");
  int rc = hamsm_stratum_reserve(c, s, K);
  if (rc) return rc;
  Stratum &t = c->st[s];
  const int64_t nch = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  // scratch: Xb[n*D] w[n] xsq[n] closest[n] cum[n] cand[L*n] chunk_tot[nch] u[(K-1)*L] chosen[K] sub[n]
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return o;
  };
  size_t o_Xb = take(sizeof(double) * n * D), o_w = take(sizeof(double) * n), o_xs = take(sizeof(double) * n),
         o_cl = take(sizeof(double) * n), o_cum = take(sizeof(double) * n), o_cand = take(sizeof(double) * n * L),
         o_ct = take(sizeof(double) * nch), o_u = take(sizeof(double) * (size_t)(K > 1 ? (K - 1) * L : 1)),
         o_ch = take(sizeof(long long) * K), o_sub = take(sizeof(long long) * (h_sub ? n : 1));
  rc = hamsm_ensure_scratch(c, off);
  if (rc) return rc;
  char *b = (char *)c->scratch;
  if (K > 1) HIPCHK(c, hipMemcpyAsync(b + o_u, h_u, sizeof(double) * (size_t)(K - 1) * L, hipMemcpyHostToDevice, c->stream));
  if (h_sub) HIPCHK(c, hipMemcpyAsync(b + o_sub, h_sub, sizeof(long long) * n, hipMemcpyHostToDevice, c->stream));
  hamsm_timer_begin(c);
  const long long *d_sub = h_sub ? (const long long *)(b + o_sub) : nullptr;
  unsigned grid = (unsigned)((n + 255) / 256);
  if (dt == HAMSM_F32)
    hipLaunchKernelGGL(k_seed_gather<float>, dim3(grid), dim3(256), 0, c->stream, (const float *)X, ldx, D, rows, d_sub, n, wf,
                       (double *)(b + o_Xb), (double *)(b + o_w), (double *)(b + o_xs));
  else
    hipLaunchKernelGGL(k_seed_gather<double>, dim3(grid), dim3(256), 0, c->stream, (const double *)X, ldx, D, rows, d_sub, n,
                       wf, (double *)(b + o_Xb), (double *)(b + o_w), (double *)(b + o_xs));
  size_t lds = sizeof(double) * ((size_t)SEED_LMAX * D + (size_t)SEED_LMAX * SEED_T + SEED_LMAX);
  HIPCHK(c, hipFuncSetAttribute((const void *)k_seed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_seed, dim3(1), dim3(SEED_T), lds, c->stream, (const double *)(b + o_Xb), (const double *)(b + o_w),
                     (const double *)(b + o_xs), n, D, K, (long long)first, (const double *)(b + o_u), L,
                     (double *)(b + o_cl), (double *)(b + o_cand), (double *)(b + o_cum), (double *)(b + o_ct), t.C,
                     (long long *)(b + o_ch));
  HIPCHK(c, hipGetLastError());
  t.K = K;
  c->table_dirty = true;
  HIPCHK(c, hipMemsetAsync(t.counts, 0, sizeof(double) * K, c->stream));  // _kmeans.py:2271
  rc = hamsm_pack_stratum(c, s, nullptr);
  hamsm_timer_end(c, ST_SEED);
  if (rc) return rc;
  if (h_chosen) HIPCHK(c, hipMemcpyAsync(h_chosen, b + o_ch, sizeof(long long) * K, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // host draw buffers are the caller's
  return HAMSM_OK;
}

// ---- minibatch update: one wave per cluster, lanes = features, samples in ascending order ---------
// _k_means_minibatch.pyx:59-108
template <typename XT>
__global__ __launch_bounds__(256) void k_update(const XT *__restrict__ X, int64_t ldx, int D, int KS,
                                                const int32_t *__restrict__ rows, int64_t n_b,
                                                const double *__restrict__ wf, const int32_t *__restrict__ lab, int K,
                                                double *C, double *counts, double *csq, double *Bpack) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= K) return;
  // pass 1: wsum in sample order (pyx:80-85)
  double wsum = 0.0;
  bool any = false;
  for (int64_t base = 0; base < n_b; base += 64) {
    int64_t i = base + lane;
    bool mt = (i < n_b) && (lab[i] == j);
    unsigned long long mask = __ballot(mt);
    any |= mask != 0;
    while (mask) {
      int bpos = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      int64_t ii = base + bpos;
      int64_t f = rows ? (int64_t)rows[ii] : ii;
      wsum = wsum + (wf ? wf[f] : 1.0);
    }
  }
  if (!(wsum > 0)) return;  // pyx:87 / :105-108: centre unchanged
  // pass 2: c = c*count; c += x*w in sample order; count += wsum; c *= 1/count  (pyx:89-104)
  const double cnt0 = counts[j];
  double acc[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    int k = lane + 64 * h;
    acc[h] = (k < D) ? C[(size_t)j * D + k] * cnt0 : 0.0;
  }
  for (int64_t base = 0; base < n_b; base += 64) {
    int64_t i = base + lane;
    bool mt = (i < n_b) && (lab[i] == j);
    unsigned long long mask = __ballot(mt);
    while (mask) {
      int bpos = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      int64_t ii = base + bpos;
      int64_t f = rows ? (int64_t)rows[ii] : ii;
      double wi = wf ? wf[f] : 1.0;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        int k = lane + 64 * h;
        if (k < D) acc[h] = acc[h] + (double)X[f * ldx + k] * wi;
      }
    }
  }
  const double cnt1 = cnt0 + wsum;
  const double alpha = 1.0 / cnt1;
  double v[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    int k = lane + 64 * h;
    v[h] = acc[h] * alpha;
    if (k < D) {
      C[(size_t)j * D + k] = v[h];
      Bpack[((size_t)(j >> 4) * KS + (k >> 2)) * 64 + (k & 3) * 16 + (j & 15)] = v[h];
    }
  }
  if (lane == 0) counts[j] = cnt1;
  // ||c||^2 as an fma chain over ascending k (oracle sqnorm_chain): lane values broadcast in order
  double a = 0.0;
  for (int k = 0; k < D; k++) {
    double vk = __shfl(k < 64 ? v[0] : v[1], k & 63, 64);
    a = __builtin_fma(vk, vk, a);
  }
  if (lane == 0) csq[j] = a;
}

extern "C" int hamsm_minibatch_step(hamsm_ctx *c, int s, const void *X, int dt, int64_t ldx, const int32_t *rows, int64_t n_b,
                                    const double *wf, int32_t *labels_out) {
  if (c->st.empty() || s < 0 || s >= c->S) return hamsm_fail(c, HAMSM_EINVAL, "bad stratum");
  Stratum &t = c->st[s];
  if (t.K == 0) return hamsm_fail(c, HAMSM_EUNFIT, "stratum %d has no centroids", s);
  if (n_b <= 0) return HAMSM_OK;
  int32_t *lab = labels_out;
  if (!lab) {
    // labels scratch lives at the end of the scratch arena so that it does not alias seed buffers in flight
    int rc = hamsm_ensure_scratch(c, sizeof(int32_t) * (size_t)n_b + 256);
    if (rc) return rc;
    lab = (int32_t *)c->scratch;
  }
  hamsm_timer_begin(c);
  int rc = hamsm_assign_rows(c, s, X, dt, ldx, rows, n_b, lab);
  if (rc) return rc;
  unsigned grid = (unsigned)((t.K + 3) / 4);
  if (dt == HAMSM_F32)
    hipLaunchKernelGGL(k_update<float>, dim3(grid), dim3(256), 0, c->stream, (const float *)X, ldx, c->D, c->KS, rows, n_b, wf,
                       lab, t.K, t.C, t.counts, t.csq, t.Bpack);
  else
    hipLaunchKernelGGL(k_update<double>, dim3(grid), dim3(256), 0, c->stream, (const double *)X, ldx, c->D, c->KS, rows, n_b,
                       wf, lab, t.K, t.C, t.counts, t.csq, t.Bpack);
  hamsm_timer_end(c, ST_STEP);
  HIPCHK(c, hipGetLastError());
  return HAMSM_OK;
}

// ---- reassignment ---------------------------------------------------------------------------------
template <typename XT>
__global__ void k_reassign(const XT *__restrict__ X, int64_t ldx, int D, int KS, const int32_t *__restrict__ rows,
                           const int32_t *__restrict__ cl, const long long *__restrict__ brow, int n_re, double new_count,
                           double *C, double *counts, double *csq, double *Bpack) {
  int i = blockIdx.x;
  if (i >= n_re) return;
  const int j = cl[i];
  if (brow) {
    int64_t r = brow[i];
    int64_t f = rows ? (int64_t)rows[r] : r;
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
      double v = (double)X[f * ldx + k];
      C[(size_t)j * D + k] = v;  // _kmeans.py:1667
      Bpack[((size_t)(j >> 4) * KS + (k >> 2)) * 64 + (k & 3) * 16 + (j & 15)] = v;
    }
    if (threadIdx.x == 0) {
      double a = 0.0;
      for (int k = 0; k < D; k++) {
        double v = (double)X[f * ldx + k];
        a = __builtin_fma(v, v, a);
      }
      csq[j] = a;
    }
  }
  if (threadIdx.x == 0) counts[j] = new_count;  // :1672
}

extern "C" int hamsm_reassign(hamsm_ctx *c, int s, const void *X, int dt, int64_t ldx, const int32_t *rows,
                              const int32_t *h_cl, const int64_t *h_brow, int n_re, double new_count) {
  if (c->st.empty() || s < 0 || s >= c->S) return hamsm_fail(c, HAMSM_EINVAL, "bad stratum");
  if (n_re <= 0) return HAMSM_OK;
  Stratum &t = c->st[s];
  for (int i = 0; i < n_re; i++)
    if (h_cl[i] < 0 || h_cl[i] >= t.K) return hamsm_fail(c, HAMSM_EINVAL, "reassign cluster out of range");
  size_t o_b = (sizeof(int32_t) * n_re + 255) & ~(size_t)255;
  int rc = hamsm_ensure_scratch(c, o_b + sizeof(long long) * n_re);
  if (rc) return rc;
  char *b = (char *)c->scratch;
  HIPCHK(c, hipMemcpyAsync(b, h_cl, sizeof(int32_t) * n_re, hipMemcpyHostToDevice, c->stream));
  if (h_brow) HIPCHK(c, hipMemcpyAsync(b + o_b, h_brow, sizeof(long long) * n_re, hipMemcpyHostToDevice, c->stream));
  const long long *d_brow = h_brow ? (const long long *)(b + o_b) : nullptr;
  hamsm_timer_begin(c);
  if (dt == HAMSM_F32)
    hipLaunchKernelGGL(k_reassign<float>, dim3(n_re), dim3(64), 0, c->stream, (const float *)X, ldx, c->D, c->KS, rows,
                       (const int32_t *)b, d_brow, n_re, new_count, t.C, t.counts, t.csq, t.Bpack);
  else
    hipLaunchKernelGGL(k_reassign<double>, dim3(n_re), dim3(64), 0, c->stream, (const double *)X, ldx, c->D, c->KS, rows,
                       (const int32_t *)b, d_brow, n_re, new_count, t.C, t.counts, t.csq, t.Bpack);
  hamsm_timer_end(c, ST_REASSIGN);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HAMSM_OK;
}

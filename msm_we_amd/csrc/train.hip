// train.hip -- a5/a6/a7: stratified mini-batch k-means training on device-resident frames.
//   hamsm_train_prepass / hamsm_train_build_rows[_batched] : planning support for
//       do_stratified_clustering (reference _clustering.py:843-899, quirks Q1/Q2)
//   hamsm_kmeans_seed[_multi]  : k-means++ (sklearn _kmeans.py:214-276), one workgroup per stratum
//   hamsm_minibatch_step / _run: _labels_inertia + _minibatch_update_dense
//                                (_kmeans.py:1620-1642, _k_means_minibatch.pyx:59-108)
//   hamsm_reassign             : low-count reassignment (_kmeans.py:1652-1673)
// These kernels work on 10^3-row batches: they are latency-bound, so they are shaped for a single
// round trip to L2 per phase (all operand loads issued before the first use) and for many small
// workgroups, and several strata share one launch.  Every reduction follows the order fixed in
// oracle/hamsm_oracle.c so that results are bit-identical.
#include <thread>

#include "common.h"
#include <type_traits>

#include <time.h>

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

// Pointers that arrive inside a descriptor are generic to the compiler (flat_load/flat_atomic, which tie up both
// memory counters); every one of them is device memory, so the kernels address them as such.
#define GAS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ GAS T *gp(T *p) {
  return (GAS T *)p;
}

// ---- prepass: raw bin or -1 (basis/target) per frame + per-iteration histogram --------------------
__global__ void k_prepass(MapperDev m, const double *__restrict__ p, const int64_t *__restrict__ offs, int64_t n_iters,
                          int S, int32_t *__restrict__ bin, unsigned long long *__restrict__ hist, int *err, int nbt) {
  extern __shared__ unsigned int lh[];  // [S] | [nbt] boundaries of a rectilinear mapper (searched per frame: from LDS, not L2)
  float *lb = nbt > 0 ? (float *)(lh + S) : nullptr;
  for (int k = threadIdx.x; k < nbt; k += blockDim.x) lb[k] = m.bounds[k];
  for (int64_t t = blockIdx.x; t < n_iters; t += gridDim.x) {
    for (int s = threadIdx.x; s < S; s += blockDim.x) lh[s] = 0;
    __syncthreads();
    for (int64_t i = offs[t] + threadIdx.x; i < offs[t + 1]; i += blockDim.x) {
      double pi[HAMSM_MAX_NDIM];
      for (int d = 0; d < m.ndim; d++) pi[d] = p[i * m.ndim + d];
      int b;
      if (dev_in_box(m.target, m.ndim, pi) || dev_in_box(m.basis, m.ndim, pi)) b = -1;  // _clustering.py:872-874
      else {
        b = dev_bin_of_staged(m, lb, pi, i);  // :877 (no we_remap)
        if (b >= S) b = -1;
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
  CTX(c);
  (void)n;
  if (!c->have_mapper || !c->have_bounds) return hamsm_fail(c, HAMSM_EINVAL, "mapper/bounds not set");
  if (n_iters <= 0) return HAMSM_OK;
  if (int rcb = hamsm_take_frame_bins(c, "train_prepass")) return rcb;
  size_t ob = sizeof(int64_t) * (n_iters + 1), hb = sizeof(unsigned long long) * (size_t)n_iters * c->S;
  size_t o_h = (ob + 255) & ~(size_t)255;
  int rc = hamsm_ensure_scratch(c, o_h + hb);
  if (rc) return rc;
  int64_t *d_offs = (int64_t *)c->scratch;
  unsigned long long *d_hist = (unsigned long long *)((char *)c->scratch + o_h);
  HIPCHK(c, hipMemcpyAsync(d_offs, h_offs, ob, hipMemcpyHostToDevice, c->stream));
  hamsm_timer_begin(c);
  unsigned grid = (unsigned)(n_iters < 4096 ? n_iters : 4096);
  const int nbt = hamsm_mapper_stage_count(c->mapper);
  hipLaunchKernelGGL(k_prepass, dim3(grid), dim3(256), sizeof(unsigned int) * (c->S + nbt), c->stream, c->mapper, p, d_offs, n_iters,
                     c->S, bin, d_hist, c->d_err, nbt);
  hamsm_timer_end(c, ST_PREPASS);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(h_hist, d_hist, hb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return hamsm_check_device_error(c, "train_prepass");
}

// ---- row lists: one workgroup per training call, stable -------------------------------------------
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

__global__ __launch_bounds__(BR_T) void k_build_rows(const int32_t *__restrict__ bin_all, const long long *__restrict__ range,
                                                     const int32_t *__restrict__ cr_all, int S, int32_t *__restrict__ rank_all,
                                                     int32_t *__restrict__ rows_all, long long *__restrict__ counts,
                                                     long long *__restrict__ starts) {
  __shared__ int sh[BR_T];
  const int g = blockIdx.x, tid = threadIdx.x;
  const long long g0 = range[2 * g], n = range[2 * g + 1] - g0;
  const int32_t *bin = bin_all + g0, *cr = cr_all + (size_t)g * S;
  int32_t *rank = rank_all + g0, *rows = rows_all + g0;
  const long long per = (n + BR_T - 1) / BR_T, a = tid * per, b = (a + per < n) ? a + per : n;
  int v = 0;
  for (long long i = a; i < b; i++) v += bin[i] >= 0;
  int tot;
  int base = block_excl_scan(v, sh, &tot);
  for (long long i = a; i < b; i++)
    if (bin[i] >= 0) rank[i] = base++;
  __syncthreads();
  long long start = 0;
  for (int s = 0; s < S; s++) {
    int cnt = 0;
    for (long long i = a; i < b; i++) cnt += (bin[i] >= 0 && cr[bin[i]] == s);
    int ts;
    int o = block_excl_scan(cnt, sh, &ts);
    if (ts) {
      for (long long i = a; i < b; i++)
        if (bin[i] >= 0 && cr[bin[i]] == s) rows[start + o++] = rank[i];
    }
    if (tid == 0) {
      counts[(size_t)g * S + s] = ts;
      starts[(size_t)g * S + s] = g0 + start;
    }
    start += ts;
  }
}

// The same lists by ONE WAVE per call, for calls of up to BR_WAVE_MAX rows (a training call is one WE iteration: ~10^4):
// 64 rows at a time, positions from ballots (mbcnt = set bits below this lane), no barrier.  Two coalesced sweeps whatever
// the number of strata: the first counts the rows of every stratum (per-wave counters in LDS, one update per stratum present in
// a chunk), the second places them behind the strata's starts; the rank of a row among the unfiltered ones is a running count
// in both.  (The call's remap row is staged in LDS: the stratum of a row is otherwise a dependent load behind its bin.)
#define BR_WAVE_MAX 65536
#define BR_SREG 8  // up to this many strata: counters and cursors in registers (no LDS round trip per stratum and chunk)
template <bool SREG>
__global__ __launch_bounds__(256) void k_build_rows_wave(const int32_t *__restrict__ bin_all, const long long *__restrict__ range,
                                                         const int32_t *__restrict__ cr_all, int S, long long G,
                                                         int32_t *__restrict__ rows_all, long long *__restrict__ counts,
                                                         long long *__restrict__ starts) {
  extern __shared__ int br_lds[];  // [4 waves][2][S]: counters / cursors, remap row
  const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= G) return;
  const int lane = threadIdx.x & 63;
  int *wc = br_lds + (size_t)(threadIdx.x >> 6) * 2 * S, *lcr = wc + S;
  const long long g0 = range[2 * g];
  const int n = (int)(range[2 * g + 1] - g0);
  const int32_t *bin = bin_all + g0, *cr = cr_all + (size_t)g * S;
  int32_t *rows = rows_all + g0;
  for (int s = lane; s < S; s += 64) {
    wc[s] = 0;
    lcr[s] = cr[s];
  }
  auto below = [&](unsigned long long m) {  // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
  };
  constexpr int U = 8;  // chunks whose loads are in flight together (one chunk at a time exposes a memory round trip each)
  int rc[BR_SREG];      // (SREG) wave-uniform counters, then cursors
#pragma unroll
  for (int s = 0; s < BR_SREG; s++) rc[s] = 0;
  for (int c0 = 0; c0 < n; c0 += 64 * U) {
    int bb[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = c0 + u * 64 + lane;
      bb[u] = bin[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = c0 + u * 64 + lane;
      const bool v = i < n && bb[u] >= 0;
      const int sl = lcr[v ? bb[u] : 0];
      if constexpr (SREG) {
#pragma unroll
        for (int s = 0; s < BR_SREG; s++) rc[s] += __builtin_popcountll(__ballot(v && sl == s));
      } else {
        unsigned long long todo = __ballot(v);
        while (todo) {  // (wave-uniform) one counter update per stratum present in the chunk
          const int l0 = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
          const int s0 = __builtin_amdgcn_readlane(sl, l0);
          const unsigned long long ms = __ballot(v && sl == s0);
          if (lane == l0) wc[s0] += __builtin_popcountll(ms);
          todo &= ~ms;
        }
      }
    }
  }
  // counters -> cursors (exclusive running sum in stratum order), counts and starts out
  int run = 0;
  if constexpr (SREG) {
#pragma unroll
    for (int s = 0; s < BR_SREG; s++) {
      const int cnt = rc[s];
      if (lane == 0 && s < S) {
        counts[(size_t)g * S + s] = cnt;
        starts[(size_t)g * S + s] = g0 + run;
      }
      rc[s] = run;
      run += cnt;
    }
  } else {
    for (int s = 0; s < S; s++) {
      const int cnt = wc[s];  // (LDS broadcast)
      if (lane == 0) {
        counts[(size_t)g * S + s] = cnt;
        starts[(size_t)g * S + s] = g0 + run;
        wc[s] = run;
      }
      run += cnt;
    }
  }
  int base = 0;
  for (int c0 = 0; c0 < n; c0 += 64 * U) {
    int bb[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = c0 + u * 64 + lane;
      bb[u] = bin[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = c0 + u * 64 + lane;
      const bool v = i < n && bb[u] >= 0;
      const int sl = lcr[v ? bb[u] : 0];
      const unsigned long long m = __ballot(v);
      const int rk = base + below(m);  // rank among the unfiltered rows of the call (:874)
      base += __builtin_popcountll(m);
      if constexpr (SREG) {
        int pos = 0;
#pragma unroll
        for (int s = 0; s < BR_SREG; s++) {
          const unsigned long long ms = __ballot(v && sl == s);
          pos = sl == s ? rc[s] + below(ms) : pos;
          rc[s] += __builtin_popcountll(ms);
        }
        if (v) rows[pos] = rk;  // rows of a stratum in row order
      } else {
        unsigned long long todo = m;
        while (todo) {
          const int l0 = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
          const int s0 = __builtin_amdgcn_readlane(sl, l0);
          const unsigned long long ms = __ballot(v && sl == s0);
          const int cur = wc[s0];
          if (v && sl == s0) rows[cur + below(ms)] = rk;  // rows of a stratum in row order
          if (lane == l0) wc[s0] = cur + __builtin_popcountll(ms);
          todo &= ~ms;
        }
      }
    }
  }
}

extern "C" int hamsm_train_build_rows_batched(hamsm_ctx *c, const int32_t *bin, int64_t G, const int64_t *h_range,
                                              const int32_t *h_cr, int32_t *rows, int64_t *h_counts, int64_t *h_starts) {
  CTX(c);
  if (G <= 0) return HAMSM_OK;
  int64_t hi = 0, nmax = 0;
  for (int64_t g = 0; g < G; g++) {
    if (h_range[2 * g + 1] < h_range[2 * g]) return hamsm_fail(c, HAMSM_EINVAL, "bad group range");
    if (h_range[2 * g + 1] > hi) hi = h_range[2 * g + 1];
    if (h_range[2 * g + 1] - h_range[2 * g] > nmax) nmax = h_range[2 * g + 1] - h_range[2 * g];
  }
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return o;
  };
  // the calls' remap rows name strata: an entry outside [0, S) would index the kernels' per-stratum counters out of bounds
  for (int64_t i = 0; i < G * (int64_t)c->S; i++)
    if (h_cr[i] < 0 || h_cr[i] >= c->S)
      return hamsm_fail(c, HAMSM_EINVAL, "call_remap[%lld][%lld] = %d is not a stratum (0..%d)", (long long)(i / c->S), (long long)(i % c->S), h_cr[i], c->S - 1);
  size_t o_rg = take(sizeof(long long) * 2 * G), o_cr = take(sizeof(int32_t) * G * c->S),
         o_cnt = take(sizeof(long long) * G * c->S), o_st = take(sizeof(long long) * G * c->S),
         o_rank = take(sizeof(int32_t) * (size_t)hi);
  int rc = hamsm_ensure_scratch(c, off);
  if (rc) return rc;
  char *b = (char *)c->scratch;
  HIPCHK(c, hipMemcpyAsync(b + o_rg, h_range, sizeof(long long) * 2 * G, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_cr, h_cr, sizeof(int32_t) * G * c->S, hipMemcpyHostToDevice, c->stream));
  hamsm_timer_begin(c);
  // (the wave form keeps 32 B of LDS per stratum: more than the default 64 KB from 2049 strata on -- a 50 x 50 bin mapper -- and
  // within the CU's 160 KB for every S this library accepts, HAMSM_MAX_STRATA = 4096; beyond that the sweep form, which needs none)
  const size_t br_lds = sizeof(int) * 8 * (size_t)c->S;
  if (nmax <= BR_WAVE_MAX && br_lds <= 160 * 1024) {
    auto kern = c->S <= BR_SREG ? k_build_rows_wave<true> : k_build_rows_wave<false>;
    if (br_lds > 48 * 1024) HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)br_lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((G + 3) / 4)), dim3(256), br_lds, c->stream, bin, (const long long *)(b + o_rg),
                       (const int32_t *)(b + o_cr), c->S, (long long)G, rows, (long long *)(b + o_cnt), (long long *)(b + o_st));
  } else
    hipLaunchKernelGGL(k_build_rows, dim3((unsigned)G), dim3(BR_T), 0, c->stream, bin, (const long long *)(b + o_rg),
                       (const int32_t *)(b + o_cr), c->S, (int32_t *)(b + o_rank), rows, (long long *)(b + o_cnt),
                       (long long *)(b + o_st));
  hamsm_timer_end(c, ST_PREPASS);
  HIPCHK(c, hipGetLastError());
  // (ABI 9) both null: the caller knows the counts -- they are the prepass histogram pushed through the calls' remaps -- and
  // the lists' starts (running sums over the strata): nothing is read back, the call returns with the kernel enqueued
  if (!h_counts && !h_starts) return HAMSM_OK;
  if (!h_counts || !h_starts) return hamsm_fail(c, HAMSM_EINVAL, "train_build_rows_batched: counts and starts come together");
  HIPCHK(c, hipMemcpyAsync(h_counts, b + o_cnt, sizeof(int64_t) * G * c->S, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_starts, b + o_st, sizeof(int64_t) * G * c->S, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HAMSM_OK;
}

extern "C" int hamsm_train_build_rows(hamsm_ctx *c, const int32_t *bin, int64_t n, const int32_t *h_cr, int32_t *rows,
                                      int64_t *h_counts, int64_t *h_starts) {
  CTX(c);
  if (n <= 0) {
    for (int s = 0; s < c->S; s++) h_counts[s] = h_starts[s] = 0;
    return HAMSM_OK;
  }
  int64_t range[2] = {0, n};
  return hamsm_train_build_rows_batched(c, bin, 1, range, h_cr, rows, h_counts, h_starts);
}

// ---- k-means++ ------------------------------------------------------------------------------------
#define SEED_T 1024          // = POT_LANES of the oracle
#define SEED_GROUP 32        // = POT_GROUP
#define SEED_LMAX 16
static_assert(SEED_T == POT_LANES, "potential partials are one per thread");

struct SeedItem {
  const void *X;        // frame 0 of this item
  const int32_t *rows;  // nullable
  const long long *sub; // nullable
  const double *wf;     // nullable, indexed like X
  const double *u;      // [(K-1)*L]
  long long n, first;
  int K, L;
  void *XbT;            // [D][n]  feature-major copy of the batch (coalesced over rows): double, or the frames' type
  double *w, *xsq, *closest, *cum, *cand /*[L][n]*/, *chunk_tot;
  double *G;            // k_seed_g: [n][n] Gram matrix of the batch; k_seed: null, or [L][D] candidate rows that do not fit LDS
  double *C;            // stratum centroids [K][D]
  long long *chosen;    // [K]
};

// BT: element type of the feature-major copy (XT for the on-chip kernel, double for the general one)
template <typename XT, typename BT>
__global__ void k_seed_gather(const SeedItem *__restrict__ items, int64_t ldx, int D) {
  const SeedItem it = items[blockIdx.y];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= it.n) return;
  const XT *X = (const XT *)it.X;
  int64_t r = it.sub ? it.sub[i] : i;
  int64_t f = it.rows ? (int64_t)it.rows[r] : r;
  double a = 0.0;
  for (int k = 0; k < D; k++) {
    double v = (double)X[f * ldx + k];
    ((BT *)it.XbT)[(size_t)k * it.n + i] = (BT)v;  // (exact: BT is double or the type v came from)
    a = __builtin_fma(v, v, a);  // row_norms(X, squared=True) restated as an fma chain (oracle sqnorm_chain)
  }
  it.xsq[i] = a;
  it.w[i] = it.wf ? it.wf[f] : 1.0;
}

// sum of the 1024 per-thread partials in the oracle's order: 32 consecutive lanes in lane order, then the
// 32 group sums in group order.  part: LDS [SEED_T]; result valid in every thread after the call.
__device__ __forceinline__ double seed_reduce(double *part, double *grp, int tid) {
  __syncthreads();
  if (tid < SEED_T / SEED_GROUP) {
    double gs = 0.0;
    for (int t = 0; t < SEED_GROUP; t++) gs = gs + part[tid * SEED_GROUP + t];
    grp[tid] = gs;
  }
  __syncthreads();
  double s = 0.0;
  for (int g = 0; g < SEED_T / SEED_GROUP; g++) s = s + grp[g];
  return s;
}

// one workgroup per stratum runs all K-1 greedy steps
__global__ __launch_bounds__(SEED_T) void k_seed(const SeedItem *__restrict__ items, int D) {
  extern __shared__ double sm[];
  const SeedItem it = items[blockIdx.x];
  const GAS double *itu = gp(it.u);
  const long long n = it.n;
  const int K = it.K, L = it.L;
  // [L][D] candidate rows: in LDS, or -- rows too long for that (it.G set: L (D + 1057) 8 B > 160 KB, e.g. 12 candidates of 816
  // features) -- in global memory, where every thread's read of crow[l][k] is the same address for the whole workgroup (one L1 /
  // L2 transaction per wave and element: slower than LDS, never a limit)
  double *crow = it.G ? it.G : sm;
  double *part = it.G ? sm : sm + (size_t)L * D;  // [L][SEED_T] per-thread partial potentials
  double *grp = part + L * SEED_T;           // [L][SEED_T/SEED_GROUP]
  double *pots = grp + L * (SEED_T / SEED_GROUP);  // [L]
  __shared__ long long cid[SEED_LMAX];
  __shared__ double s_pot;
  __shared__ int s_best;
  const int tid = threadIdx.x;
  const double *XbT = (const double *)it.XbT;

  // first centre (_kmeans.py:225-238)
  for (int k = tid; k < D; k += SEED_T) {
    double v = XbT[(size_t)k * n + it.first];
    crow[k] = v;
    it.C[k] = v;
  }
  if (tid == 0) it.chosen[0] = it.first;
  if (it.G) __threadfence_block();
  __syncthreads();
  {
    const double xs0 = it.xsq[it.first];
    double a = 0.0;
    for (long long i = tid; i < n; i += SEED_T) {
      double acc = 0.0;
      for (int k = 0; k < D; k++) acc = __builtin_fma(crow[k], XbT[(size_t)k * n + i], acc);
      double v = ((-2.0 * acc) + xs0) + it.xsq[i];
      v = v > 0.0 ? v : 0.0;
      it.closest[i] = v;
      a = a + v * it.w[i];
    }
    part[tid] = a;
    double s = seed_reduce(part, grp, tid);
    if (tid == 0) s_pot = s;
    __syncthreads();
  }
  const long long nch = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  for (int cc = 1; cc < K; cc++) {
    // chunked cumulative sum of w*closest (:243-247 stable_cumsum): chunk totals, then chunk bases
    for (long long q = tid; q < nch; q += SEED_T) {
      double loc = 0.0;
      long long e = (q + 1) * SCAN_CHUNK < n ? (q + 1) * SCAN_CHUNK : n;
      for (long long i = q * SCAN_CHUNK; i < e; i++) loc = loc + it.w[i] * it.closest[i];
      it.chunk_tot[q] = loc;
    }
    __syncthreads();
    if (tid == 0) {
      double base = 0.0;
      for (long long q = 0; q < nch; q++) {
        double t = it.chunk_tot[q];
        it.chunk_tot[q] = base;  // becomes the chunk's base
        base = base + t;
      }
    }
    __syncthreads();
    // searchsorted(cum, u*pot, side='left') without materialising cum: last chunk whose base is < v, then
    // the running sum inside it (cum[i] = base + loc, the oracle's values)
    if (tid < L) {
      const double v = itu[(size_t)(cc - 1) * L + tid] * s_pot;  // :243
      long long lo = 0, hi = nch;  // first chunk whose END value >= v  <=> first q with base[q+1] >= v
      // chunk q ends with cum = base[q] + tot[q]; recompute per candidate chunk (monotone, so bisect on bases)
      while (hi - lo > 1) {
        long long mid = (lo + hi) >> 1;
        if (it.chunk_tot[mid] < v) lo = mid;  // base[mid] < v: the answer is not before chunk mid-? (see below)
        else hi = mid;
      }
      // candidates: chunk lo (base < v or lo == 0) -- the first index with cum >= v lies in chunk lo or is the
      // first element of a later chunk; scan forward from the start of chunk lo
      long long ans = n;
      for (long long q = lo; q < nch && ans == n; q++) {
        double loc = 0.0, base = it.chunk_tot[q];
        long long e = (q + 1) * SCAN_CHUNK < n ? (q + 1) * SCAN_CHUNK : n;
        for (long long i = q * SCAN_CHUNK; i < e; i++) {
          loc = loc + it.w[i] * it.closest[i];
          if (base + loc >= v) {
            ans = i;
            break;
          }
        }
      }
      if (ans > n - 1) ans = n - 1;  // :249 clip
      cid[tid] = ans;
    }
    __syncthreads();
    for (int e = tid; e < L * D; e += SEED_T) crow[e] = XbT[(size_t)(e % D) * n + cid[e / D]];
    if (it.G) __threadfence_block();  // (rows in global memory: written by other waves of this workgroup)
    __syncthreads();
    {
      double a[SEED_LMAX];
#pragma unroll
      for (int l = 0; l < SEED_LMAX; l++) a[l] = 0.0;
      for (long long i = tid; i < n; i += SEED_T) {
        double acc[SEED_LMAX];
#pragma unroll
        for (int l = 0; l < SEED_LMAX; l++) acc[l] = 0.0;
        for (int k0 = 0; k0 < D; k0 += 8) {
          double xk[8];
#pragma unroll
          for (int q = 0; q < 8; q++) xk[q] = XbT[(size_t)(k0 + q < D ? k0 + q : 0) * n + i];
#pragma unroll
          for (int q = 0; q < 8; q++)
            if (k0 + q < D) {
#pragma unroll
              for (int l = 0; l < SEED_LMAX; l++)
                if (l < L) acc[l] = __builtin_fma(crow[l * D + k0 + q], xk[q], acc[l]);
            }
        }
        const double ci = it.closest[i], xi2 = it.xsq[i], wi = it.w[i];
#pragma unroll
        for (int l = 0; l < SEED_LMAX; l++)
          if (l < L) {
            double v = ((-2.0 * acc[l]) + it.xsq[cid[l]]) + xi2;  // euclidean_distances order (:252-254)
            v = v > 0.0 ? v : 0.0;
            v = v < ci ? v : ci;  // :257 np.minimum
            it.cand[(size_t)l * n + i] = v;
            a[l] = a[l] + v * wi;
          }
      }
#pragma unroll
      for (int l = 0; l < SEED_LMAX; l++)
        if (l < L) part[l * SEED_T + tid] = a[l];
    }
    __syncthreads();
    // potentials of the L candidates (:258), oracle order
    for (int e = tid; e < L * (SEED_T / SEED_GROUP); e += SEED_T) {
      int l = e / (SEED_T / SEED_GROUP), g = e % (SEED_T / SEED_GROUP);
      double gs = 0.0;
      for (int t = 0; t < SEED_GROUP; t++) gs = gs + part[l * SEED_T + g * SEED_GROUP + t];
      grp[l * (SEED_T / SEED_GROUP) + g] = gs;
    }
    __syncthreads();
    if (tid < L) {
      double s = 0.0;
      for (int g = 0; g < SEED_T / SEED_GROUP; g++) s = s + grp[tid * (SEED_T / SEED_GROUP) + g];
      pots[tid] = s;
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
      it.chosen[cc] = cid[best];
    }
    __syncthreads();
    const int best = s_best;
    for (long long i = tid; i < n; i += SEED_T) it.closest[i] = it.cand[(size_t)best * n + i];
    for (int k = tid; k < D; k += SEED_T) it.C[(size_t)cc * D + k] = crow[best * D + k];
    __syncthreads();
  }
}

#ifdef SEED_PROFILE
__device__ unsigned long long g_seed_prof[16];
#define PROF_T(idx)                                                        \
  if (tid == 0 && blockIdx.x == 0) {                                       \
    unsigned long long t_ = __builtin_amdgcn_s_memtime();                  \
    g_seed_prof[idx] += t_ - t_last;                                       \
    t_last = t_;                                                           \
  }
#else
#define PROF_T(idx)
#endif
#define SEED_NLDS 3072
#define SEED_TL 512                       // threads of the on-chip kernel: 8 waves, 6 rows each lane
#define SEED_LPT (SEED_T / SEED_TL)       // canonical lanes (of the oracle's POT_LANES) per thread
#define SEED_RPL (SEED_NLDS / SEED_T)     // rows per lane: lane t owns rows t, t + 1024, t + 2048
#define SEED_RPT (SEED_LPT * SEED_RPL)    // rows per thread
#define SEED_LFAST 9                      // most candidates the on-chip kernel takes (K < 2981)

// ---- k-means++ from the batch's Gram matrix ---------------------------------------------------------
// Every distance a greedy step needs is between two ROWS OF THE BATCH, and the dot product inside it is the in-order fma
// chain over the features -- exactly what v_mfma_f64_16x16x4_f64 accumulates.  So the whole chip computes G = Xb Xb^T once
// (n <= 3072: 75 MB, a fraction of a millisecond), and the K-1 serial greedy steps of the one workgroup that owns the
// stratum shrink from L n D fmas each to L n loads of finished dot products.  Bit-identical to k_seed / the oracle.
template <typename XT>
__global__ __launch_bounds__(256) void k_seed_gram(const SeedItem *__restrict__ items, int D) {
  const SeedItem it = items[blockIdx.z];
  const int n = (int)it.n;
  const int i0 = blockIdx.y * 128, j0 = blockIdx.x * 128;
  if (i0 >= n || j0 >= n) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wi = i0 + (wv >> 1) * 64, wj = j0 + (wv & 1) * 64;
  const GAS XT *X = (const GAS XT *)gp(it.XbT);
  GAS double *G = gp(it.G);
  v4f64 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = (v4f64){0.0, 0.0, 0.0, 0.0};
  int ia[4], jb[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    int i = wi + t * 16 + (lane & 15), j = wj + t * 16 + (lane & 15);
    ia[t] = i < n ? i : n - 1;
    jb[t] = j < n ? j : n - 1;
  }
  const int kq = lane >> 4;
  for (int k0 = 0; k0 < D; k0 += 4) {
    const int k = k0 + kq;
    const size_t ko = (size_t)(k < D ? k : D - 1) * n;
    double a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      a[t] = (double)X[ko + ia[t]];
      b[t] = (double)X[ko + jb[t]];
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {  // features past D: fma(0, 0, acc) == acc
      a[t] = k < D ? a[t] : 0.0;
      b[t] = k < D ? b[t] : 0.0;
    }
#pragma unroll
    for (int ti = 0; ti < 4; ti++)
#pragma unroll
      for (int tj = 0; tj < 4; tj++) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
  }
#pragma unroll
  for (int ti = 0; ti < 4; ti++)
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = wi + ti * 16 + kq + 4 * r, col = wj + tj * 16 + (lane & 15);
        if (row < n && col < n) G[(size_t)row * n + col] = acc[ti][tj][r];
      }
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
  unsigned lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) & 0xffffffffll), l);
  unsigned hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | lo);
}

#define SEED_NCH (SEED_NLDS / SCAN_CHUNK)   // 48 chunks of the cumulative sum
#define SEED_PST (SEED_NCH + 1)             // stride of the transposed product array: lanes = chunks, conflict-free
#define SEED_PL (32 * 33)                   // one candidate's 1024 partial potentials, groups of 32 padded to 33
// LDS of k_seed_g in doubles
#define SEED_G_LDS(LF) ((size_t)(LF) * SEED_PL + (LF) * 32 + 16 + SEED_NLDS + 2 * 64 * SEED_PST + 2 * SEED_NCH)
template <typename XT, int LF>
__global__ __launch_bounds__(SEED_TL) void k_seed_g(const SeedItem *__restrict__ items, int D) {
  extern __shared__ double sm[];
  const SeedItem it = items[blockIdx.x];
  const int n = (int)it.n;
  const int K = it.K;
  constexpr int L = LF;                 // (== it.L, checked by the host)
  double *part = sm;                    // [LF][SEED_PL] per-lane partial potentials (oracle lane t -> (t / 32) * 33 + t % 32)
  double *grp = part + LF * SEED_PL;    // [LF][32]
  double *pots = grp + LF * 32;         // [16]
  double *lxsq = pots + 16;             // [SEED_NLDS] ||x||^2
  double *pT = lxsq + SEED_NLDS;        // [64][SEED_PST] w_i * closest_i, row i at (i % 64, i / 64); 0 past the batch
  double *lbase = pT + 64 * SEED_PST;   // [SEED_NCH] chunk totals
  double *pC = lbase + SEED_NCH;        // [64][SEED_PST] running sum of the products inside each chunk
  double *lexc = pC + 64 * SEED_PST;    // [SEED_NCH] exclusive running sum of the chunk totals (the chunks' bases)
  __shared__ int cid[SEED_LFAST];
  __shared__ double s_pot;
  __shared__ int s_best;
  const int tid = threadIdx.x;
  const GAS XT *XbT = (const GAS XT *)gp(it.XbT);
  const GAS double *G = gp(it.G);
  const GAS double *itu = gp(it.u);
  GAS double *itC = gp(it.C);
  GAS long long *itchosen = gp(it.chosen);
  // row rr of this thread: oracle lane tid + (rr / RPL) * TL, row lane + (rr % RPL) * 1024
  auto row_of = [&](int rr) { return tid + (rr / SEED_RPL) * SEED_TL + (rr % SEED_RPL) * SEED_T; };
  auto pidx = [&](int i) { return (i & (SCAN_CHUNK - 1)) * SEED_PST + (i >> 6); };
  static_assert(SCAN_CHUNK == 64, "pidx");
  // this thread's rows stay in registers for the whole seeding: index (clamped), weight (+0 past the batch: a partial sum
  // plus v * 0.0 is the partial sum), ||x||^2, closest
  int ii[SEED_RPT];
  double wr[SEED_RPT], x2[SEED_RPT], cl[SEED_RPT];
#pragma unroll
  for (int r = 0; r < SEED_RPT; r++) {
    const int i = row_of(r);
    ii[r] = i < n ? i : n - 1;
    wr[r] = i < n ? gp(it.w)[ii[r]] : 0.0;
    x2[r] = gp(it.xsq)[ii[r]];
  }
  for (int i = tid; i < SEED_NLDS; i += SEED_TL) lxsq[i] = i < n ? gp(it.xsq)[i] : 0.0;
  for (int e = tid; e < 64 * SEED_PST + SEED_NCH; e += SEED_TL) pT[e] = 0.0;  // (and lbase)
  for (int k = tid; k < D; k += SEED_TL) itC[k] = (double)XbT[(size_t)k * n + it.first];
  if (tid == 0) itchosen[0] = it.first;
  __syncthreads();
  // sum of nl candidates' 1024 partials in the oracle's order: 32 consecutive lanes, then the 32 group sums
  auto reduce1 = [&](int nl) {
    __syncthreads();
    if (tid < nl * 32) {
      const int l = tid >> 5, g = tid & 31;
      double pv[32];
#pragma unroll
      for (int t = 0; t < 32; t++) pv[t] = part[l * SEED_PL + g * 33 + t];
      double gs = 0.0;
#pragma unroll
      for (int t = 0; t < 32; t++) gs = gs + pv[t];
      grp[l * 32 + g] = gs;
    }
    __syncthreads();
  };
  {
    const double xs0 = lxsq[it.first];
    double g0[SEED_RPT];
#pragma unroll
    for (int r = 0; r < SEED_RPT; r++) g0[r] = G[(size_t)it.first * n + ii[r]];
#pragma unroll
    for (int lp = 0; lp < SEED_LPT; lp++) {
      double a = 0.0;
#pragma unroll
      for (int r = 0; r < SEED_RPL; r++) {
        const int rr = lp * SEED_RPL + r;
        double v = ((-2.0 * g0[rr]) + xs0) + x2[rr];
        v = v > 0.0 ? v : 0.0;
        cl[rr] = v;
        const double pr = v * wr[rr];
        if (row_of(rr) < n) pT[pidx(row_of(rr))] = pr;
        a = a + pr;
      }
      const int ln = tid + lp * SEED_TL;
      part[(ln >> 5) * 33 + (ln & 31)] = a;
    }
    reduce1(1);
    if (tid == 0) {
      double s = 0.0;
      for (int g = 0; g < 32; g++) s = s + grp[g];
      s_pot = s;
    }
    __syncthreads();
  }
  const int nch = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  const int lane = tid & 63, wv = tid >> 6;
  static_assert(SEED_LFAST <= 2 * (SEED_TL / 64), "two candidates per wave");
  double un0 = (K > 1 && wv < L) ? itu[wv] : 0.0, un1 = (K > 1 && wv + SEED_TL / 64 < L) ? itu[wv + SEED_TL / 64] : 0.0;
#ifdef SEED_PROFILE
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
  for (int cc = 1; cc < K; cc++) {
    if (tid < SEED_NCH) {  // per chunk: the running sum of its 64 products in row order (zeros past the batch)
      double pv[SCAN_CHUNK];
#pragma unroll
      for (int z = 0; z < SCAN_CHUNK; z++) pv[z] = pT[z * SEED_PST + tid];
      double loc = 0.0;
#pragma unroll
      for (int z = 0; z < SCAN_CHUNK; z++) {
        loc = loc + pv[z];
        pC[z * SEED_PST + tid] = loc;
      }
      lbase[tid] = loc;
      // the chunks' bases: the exclusive running sum of the totals in chunk order (a serial left fold: the oracle's order), formed
      // HERE, once, by the wave that holds the totals in its lanes -- every total goes round as a scalar (v_readlane), lane q keeps
      // the sum in front of chunk q.  (Round 5: every searching wave used to form it again from 48 LDS broadcast reads.)
      double run = 0.0, mine = 0.0;
#pragma unroll
      for (int q = 0; q < SEED_NCH; q++) {
        const double tq = readlane_f64(loc, q);
        mine = tid == q ? run : mine;
        run = run + tq;
      }
      lexc[tid] = mine;
    }
    __syncthreads();
    PROF_T(0)
    PROF_T(1)
    if (wv < L) {  // wave wv draws candidates wv and wv + 8
      const double u0 = un0, u1 = un1;
      if (cc + 1 < K) {  // next step's draws: their latency hides behind this step
        un0 = itu[(size_t)cc * L + wv];
        if (wv + SEED_TL / 64 < L) un1 = itu[(size_t)cc * L + wv + SEED_TL / 64];
      }
      // lane q keeps the base of chunk q (the exclusive running sum of the chunk totals, formed once in the chunk phase)
      const double mybase = lane < SEED_NCH ? lexc[lane] : 0.0;
      for (int l = wv; l < L; l += SEED_TL / 64) {
        const double v = (l == wv ? u0 : u1) * s_pot;  // :243
        // searchsorted over the cumulative sum: the last chunk whose base is below v (what a bisection of the
        // non-decreasing bases returns), then the first row of it whose cumulative value reaches v
        const unsigned long long below = __ballot(lane < nch && mybase < v);
        const int lo = below ? 63 - __builtin_clzll(below) : 0;
        double base = readlane_f64(mybase, lo);
        int ans = n;
        for (int q = lo; q < nch && ans == n; q++) {
          const int e = n - q * SCAN_CHUNK;  // rows of this chunk (>= 64 except in the last one)
          const double loc = pC[lane * SEED_PST + q];
          const unsigned long long hit = __ballot(lane < e && base + loc >= v);
          if (hit) ans = q * SCAN_CHUNK + __builtin_ctzll(hit);
          base = base + pC[(SCAN_CHUNK - 1) * SEED_PST + q];  // the chunk's total: the next chunk's base
        }
        if (ans > n - 1) ans = n - 1;  // :249
        if (lane == 0) cid[l] = ans;
      }
    }
    __syncthreads();
    PROF_T(2)
    double cnd[SEED_RPT][LF];
    {
      size_t go[LF];
      double xc[LF];
#pragma unroll
      for (int l = 0; l < LF; l++) {
        go[l] = (size_t)cid[l] * n;
        xc[l] = lxsq[cid[l]];
      }
#pragma unroll
      for (int r = 0; r < SEED_RPT; r++)
#pragma unroll
        for (int l = 0; l < LF; l++) cnd[r][l] = G[go[l] + ii[r]];
      PROF_T(3)
#pragma unroll
      for (int lp = 0; lp < SEED_LPT; lp++) {
        double a[LF];
#pragma unroll
        for (int l = 0; l < LF; l++) a[l] = 0.0;
#pragma unroll
        for (int r = 0; r < SEED_RPL; r++) {
          const int rr = lp * SEED_RPL + r;
#pragma unroll
          for (int l = 0; l < LF; l++) {
            double v = ((-2.0 * cnd[rr][l]) + xc[l]) + x2[rr];  // :252-254
            v = v > 0.0 ? v : 0.0;
            v = v < cl[rr] ? v : cl[rr];  // :257
            cnd[rr][l] = v;
            a[l] = a[l] + v * wr[rr];
          }
        }
        const int ln = tid + lp * SEED_TL;
#pragma unroll
        for (int l = 0; l < LF; l++) part[l * SEED_PL + (ln >> 5) * 33 + (ln & 31)] = a[l];
      }
    }
    PROF_T(4)
    reduce1(L);
    if (wv == 0) {  // the 32 group sums of candidate `lane` in group order, then the first minimum (:261)
      double sp = 0.0;
      if (lane < L) {
        double pv[32];
#pragma unroll
        for (int g = 0; g < 32; g++) pv[g] = grp[lane * 32 + g];
#pragma unroll
        for (int g = 0; g < 32; g++) sp = sp + pv[g];
      }
      int best = 0;
      double bp = readlane_f64(sp, 0);
#pragma unroll
      for (int l = 1; l < LF; l++) {
        const double pl = readlane_f64(sp, l);
        if (pl < bp) {
          bp = pl;
          best = l;
        }
      }
      if (lane == 0) {
        s_best = best;
        s_pot = bp;
        itchosen[cc] = cid[best];
      }
    }
    __syncthreads();
    PROF_T(5)
    const int best = s_best;
#pragma unroll
    for (int r = 0; r < SEED_RPT; r++) {
      double v = cnd[r][0];
#pragma unroll
      for (int l = 1; l < LF; l++) v = (l == best) ? cnd[r][l] : v;
      cl[r] = v;
      if (row_of(r) < n) pT[pidx(row_of(r))] = v * wr[r];
    }
    {
      const size_t bo = (size_t)cid[best];
      for (int k = tid; k < D; k += SEED_TL) itC[(size_t)cc * D + k] = (double)XbT[(size_t)k * n + bo];
    }
    __syncthreads();
    PROF_T(6)
  }
}

#ifdef SEED_PROFILE
extern "C" int hamsm_debug_seed_profile(unsigned long long *h_out) {
  return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_seed_prof), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int hamsm_kmeans_seed_multi(hamsm_ctx *c, int n_items, const int32_t *h_stratum, const void *X, int dt, int64_t ldx,
                                       const int64_t *h_x_off, const int32_t *d_rows, const int64_t *h_rows_off,
                                       const int64_t *h_n_b, const double *d_w, const int64_t *h_sub, const int64_t *h_n_sub,
                                       const int32_t *h_K, const int64_t *h_first, const double *h_u, const int32_t *h_L) {
  CTX(c);
  if (n_items <= 0) return HAMSM_OK;
  if (c->st.empty()) return hamsm_fail(c, HAMSM_EINVAL, "set_geometry first");
  const int D = c->D;
  const size_t esz = dt == HAMSM_F32 ? 4 : 8;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return o;
  };
  struct Lay {
    size_t XbT, w, xsq, cl, cum, cand, ct, u, ch, sub, crow;
    int64_t n;
  };
  std::vector<Lay> lay(n_items);
  size_t o_items = take(sizeof(SeedItem) * n_items);
  int64_t nmax = 0;
  for (int i = 0; i < n_items; i++) {
    int s = h_stratum[i];
    if (s < 0 || s >= c->S) return hamsm_fail(c, HAMSM_EINVAL, "bad stratum");
    for (int j = 0; j < i; j++)
      if (h_stratum[j] == s) return hamsm_fail(c, HAMSM_EINVAL, "stratum %d seeded twice in one call", s);
    int64_t n = h_n_sub[i] ? h_n_sub[i] : h_n_b[i];
    int K = h_K[i], L = h_L[i];
    if (K < 1 || n < 1 || h_first[i] < 0 || h_first[i] >= n || L < 1 || L > SEED_LMAX)
      return hamsm_fail(c, HAMSM_EINVAL, "bad seed args for stratum %d", s);
    int rc = hamsm_stratum_reserve(c, s, K);
    if (rc) return rc;
    Lay &l = lay[i];
    l.n = n;
    if (n > nmax) nmax = n;
    int64_t nch = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    l.XbT = take(sizeof(double) * n * D);
    l.w = take(sizeof(double) * n);
    l.xsq = take(sizeof(double) * n);
    l.cl = take(sizeof(double) * n);
    l.cum = take(sizeof(double) * 1);
    l.cand = take(sizeof(double) * n * L);
    l.ct = take(sizeof(double) * nch);
    l.u = take(sizeof(double) * (size_t)(K > 1 ? (K - 1) * L : 1));
    l.ch = take(sizeof(long long) * K);
    l.sub = take(sizeof(long long) * (h_n_sub[i] ? h_n_sub[i] : 1));
  }
  // batches that fit on chip, one candidate count per launch: Gram-matrix seeding (k_seed_gram + k_seed_g)
  bool fast = true;
  for (int i = 0; i < n_items; i++) fast = fast && lay[i].n <= SEED_NLDS && h_L[i] <= SEED_LFAST && h_L[i] == h_L[0];
  const size_t lds_fast = sizeof(double) * SEED_G_LDS(h_L[0]);
  // the general kernel keeps its candidate rows in LDS while they fit beside the partial potentials, else in global memory
  int Lmax = 1;
  for (int i = 0; i < n_items; i++) Lmax = h_L[i] > Lmax ? h_L[i] : Lmax;
  const size_t lds_part = sizeof(double) * (size_t)Lmax * (SEED_T + SEED_T / SEED_GROUP + 1);
  const bool crow_global = !fast && lds_part + sizeof(double) * (size_t)Lmax * D > 160 * 1024 - 256;
  if (crow_global)
    for (int i = 0; i < n_items; i++) lay[i].crow = take(sizeof(double) * (size_t)h_L[i] * D);
  const size_t g_bytes = (sizeof(double) * (size_t)nmax * nmax + 255) & ~(size_t)255;
  int g_slots = 1;  // strata whose Gram matrices are resident at once (<= ctx seed_g_budget bytes)
  size_t o_G = 0;
  if (fast) {
    g_slots = (int)(c->seed_g_budget / g_bytes);
    g_slots = g_slots < 1 ? 1 : (g_slots > n_items ? n_items : g_slots);
    o_G = take(g_bytes * g_slots);
  }
  int rc = hamsm_ensure_scratch(c, off);
  if (rc) return rc;
  char *b = (char *)c->scratch;
  std::vector<SeedItem> items(n_items);
  const double *up = h_u;
  const int64_t *sp = h_sub;
  for (int i = 0; i < n_items; i++) {
    const Lay &l = lay[i];
    int K = h_K[i], L = h_L[i];
    SeedItem &it = items[i];
    it.X = (const char *)X + (size_t)h_x_off[i] * ldx * esz;
    it.rows = d_rows ? d_rows + h_rows_off[i] : nullptr;
    it.sub = h_n_sub[i] ? (const long long *)(b + l.sub) : nullptr;
    it.wf = d_w ? d_w + h_x_off[i] : nullptr;
    it.u = (const double *)(b + l.u);
    it.n = l.n;
    it.first = h_first[i];
    it.K = K;
    it.L = L;
    it.XbT = (double *)(b + l.XbT);
    it.w = (double *)(b + l.w);
    it.xsq = (double *)(b + l.xsq);
    it.closest = (double *)(b + l.cl);
    it.cum = (double *)(b + l.cum);
    it.cand = (double *)(b + l.cand);
    it.chunk_tot = (double *)(b + l.ct);
    it.G = fast ? (double *)(b + o_G + g_bytes * (size_t)(i % g_slots)) : (crow_global ? (double *)(b + l.crow) : nullptr);
    it.C = c->st[h_stratum[i]].C;
    it.chosen = (long long *)(b + l.ch);
    if (K > 1) HIPCHK(c, hipMemcpyAsync(b + l.u, up, sizeof(double) * (size_t)(K - 1) * L, hipMemcpyHostToDevice, c->stream));
    if (h_n_sub[i]) HIPCHK(c, hipMemcpyAsync(b + l.sub, sp, sizeof(long long) * h_n_sub[i], hipMemcpyHostToDevice, c->stream));
    up += (size_t)(K - 1) * L;
    sp += h_n_sub[i];
  }
  HIPCHK(c, hipMemcpyAsync(b + o_items, items.data(), sizeof(SeedItem) * n_items, hipMemcpyHostToDevice, c->stream));
  hamsm_timer_begin(c);
  dim3 gg((unsigned)((nmax + 255) / 256), (unsigned)n_items);
  const SeedItem *d_items = (const SeedItem *)(b + o_items);
  if (dt == HAMSM_F32 && fast) hipLaunchKernelGGL((k_seed_gather<float, float>), gg, dim3(256), 0, c->stream, d_items, ldx, D);
  else if (dt == HAMSM_F32) hipLaunchKernelGGL((k_seed_gather<float, double>), gg, dim3(256), 0, c->stream, d_items, ldx, D);
  else hipLaunchKernelGGL((k_seed_gather<double, double>), gg, dim3(256), 0, c->stream, d_items, ldx, D);
  if (fast) {
    const void *fn = nullptr;
#define SEED_CASE(LV)                                                                            \
  case LV:                                                                                       \
    fn = dt == HAMSM_F32 ? (const void *)k_seed_g<float, LV> : (const void *)k_seed_g<double, LV>; \
    break;
    switch (h_L[0]) {
      SEED_CASE(1) SEED_CASE(2) SEED_CASE(3) SEED_CASE(4) SEED_CASE(5) SEED_CASE(6) SEED_CASE(7) SEED_CASE(8) SEED_CASE(9)
    }
#undef SEED_CASE
    HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fast));
    // groups of strata share the Gram slots: the stream orders a group's Gram + greedy kernels before the next group's
    for (int g0 = 0; g0 < n_items; g0 += g_slots) {
      const int ng = n_items - g0 < g_slots ? n_items - g0 : g_slots;
      const SeedItem *gi = d_items + g0;
      dim3 gm((unsigned)((nmax + 127) / 128), (unsigned)((nmax + 127) / 128), (unsigned)ng);
      if (dt == HAMSM_F32) hipLaunchKernelGGL((k_seed_gram<float>), gm, dim3(256), 0, c->stream, gi, D);
      else hipLaunchKernelGGL((k_seed_gram<double>), gm, dim3(256), 0, c->stream, gi, D);
      void *args[] = {(void *)&gi, (void *)&D};
      HIPCHK(c, hipLaunchKernel(fn, dim3((unsigned)ng), dim3(SEED_TL), args, lds_fast, c->stream));
    }
  } else {
    const size_t lds = crow_global ? lds_part : lds_part + sizeof(double) * (size_t)Lmax * D;  // (L <= 16: lds_part <= 135 KB)
    HIPCHK(c, hipFuncSetAttribute((const void *)k_seed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_seed, dim3((unsigned)n_items), dim3(SEED_T), lds, c->stream, (const SeedItem *)(b + o_items), D);
  }
  HIPCHK(c, hipGetLastError());
  {
    std::vector<int> seeded((size_t)n_items);
    for (int i = 0; i < n_items; i++) {
      seeded[i] = h_stratum[i];
      c->st[h_stratum[i]].K = h_K[i];
    }
    rc = hamsm_pack_seeded(c, seeded.data(), n_items);  // counts := 0 (_kmeans.py:2271), csq, fragments: one launch
    if (rc) return rc;
  }
  c->table_dirty = true;
  hamsm_timer_end(c, ST_SEED);
  // h_u / h_sub were pageable host memory: hipMemcpyAsync staged them before returning, no sync needed.
  c->seed_chosen_epoch = c->scratch_epoch;  // (the records below point into the scratch arena: valid until its next user)
  c->seed_chosen_dev = b + lay[0].ch;  // hamsm_kmeans_seed reads item 0's choices
  c->seed_chosen_items.resize(n_items);
  for (int i = 0; i < n_items; i++) c->seed_chosen_items[i] = {b + lay[i].ch, h_K[i]};  // (hamsm_kmeans_seed_chosen; valid until the scratch arena is reused)
  return HAMSM_OK;
}

extern "C" int hamsm_kmeans_seed_chosen(hamsm_ctx *c, int item, int64_t *h_chosen, int64_t capacity) {
  CTX(c);
  if (item < 0 || (size_t)item >= c->seed_chosen_items.size() || !h_chosen) return hamsm_fail(c, HAMSM_EINVAL, "seed_chosen: no such item of the last seeding call");
  // (the records point into the scratch arena -- gone once another call has taken it -- and the caller's buffer is the
  // caller's guess of K)
  if (c->seed_chosen_epoch != c->scratch_epoch) return hamsm_fail(c, HAMSM_EINVAL, "seed_chosen: the scratch arena has been reused since the seeding call");
  if (capacity < (int64_t)c->seed_chosen_items[item].second)
    return hamsm_fail(c, HAMSM_EINVAL, "seed_chosen: item %d has %d centroids, the buffer holds %lld", item, c->seed_chosen_items[item].second, (long long)capacity);
  HIPCHK(c, hipMemcpyAsync(h_chosen, c->seed_chosen_items[item].first, sizeof(long long) * c->seed_chosen_items[item].second, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HAMSM_OK;
}

extern "C" int hamsm_kmeans_seed(hamsm_ctx *c, int s, const void *X, int dt, int64_t ldx, const int32_t *rows, int64_t n_b,
                                 const double *wf, const int64_t *h_sub, int64_t n_sub, int K, int64_t first,
                                 const double *h_u, int L, int64_t *h_chosen) {
  CTX(c);
  int32_t st = s, k = K, l = L;
  int64_t zero = 0, nb = n_b, ns = h_sub ? n_sub : 0, f = first;
  int rc = hamsm_kmeans_seed_multi(c, 1, &st, X, dt, ldx, &zero, rows, &zero, &nb, wf, h_sub, &ns, &k, &f, h_u, &l);
  if (rc) return rc;
  if (h_chosen) HIPCHK(c, hipMemcpyAsync(h_chosen, c->seed_chosen_dev, sizeof(long long) * K, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HAMSM_OK;
}

// ---- minibatch steps ------------------------------------------------------------------------------
struct StepDev {
  const void *X;         // frame 0 of the step's call
  const int32_t *rows;   // batch row list (nullable = identity)
  const double *wf;      // nullable, indexed like X
  long long n_b;
  int32_t *labels;       // [n_b] scratch
  double *Xg;            // [n_b padded to 64][4 KS] the batch rows as f64 (rows = whole 128-byte lines), written by the
                         // E-step for the M-step (measured: an f32 copy,
                         // packed or one row per 128-byte line, makes the M-step's fetch 3x slower): a small buffer that is
                         // TLB- and cache-warm on every CU, unlike the call's span of the frame array
  unsigned long long *mbits;  // [K][wstride] membership bitmaps: bit b of word w of row j <=> batch row 64 w + b is nearest
                              // to centroid j.  Set by the E-step with fire-and-forget atomic ORs, read (in sample
                              // order by construction) and cleared by the M-step.
  int wstride;                // words per centroid (a multiple of 64, >= ceil(n_b / 64))
  double *C, *counts, *csq, *Bpack;
  int K, flags;          // flags: 1 = low-count test wanted after this step, 2 = zero-count test before it,
                         //        4 = this step's E-step evaluates the low-count test of the stratum's previous step
  int32_t *report;       // [2]
  int32_t *prev_report;  // report of the stratum's previous step (flags & 4)
  const int32_t *perm;   // flags & 1: head of the permutation the host drew for this step's re-seeding (nullable)
  int n_perm, idx;       // idx = position of the step in the run
  int32_t *halt;         // the stratum's stop word: steps with idx >= *halt are not applied
  int32_t *reason;       // HAMSM_STOP_* of the stratum
};
// what the next call's E-step will gather: its lines are pulled towards this XCD's L2 while this call computes
struct StepNext {
  const void *X;
  const int32_t *rows;
  long long n_b;
};
// launch groups of at most PACK_NY strata travel in the kernel arguments: no dependent load of the descriptor
#define PACK_NY 4
struct StepPack {
  StepDev s[PACK_NY];
  StepNext nx[PACK_NY];
  int n_next;
};

#define PF_MAX 3 // prefetch loads per lane (4 lanes per row): rows of up to 12 cache lines

// STEP_PROFILE=1: entry/exit stamps only; =2: a stamp after every phase, each behind a full s_waitcnt (perturbs overlap)
#ifdef STEP_PROFILE
#define SPROF_WG 4096
__device__ unsigned long long g_step_prof[2][SPROF_WG][8];
#define SPROF_ID                                        \
  const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x; \
  unsigned long long sp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SPROF(k, slot) sp_[slot] = wall_clock64();
#define SPROF_FLUSH(k)                            \
  if (threadIdx.x == 0 && wg_ < SPROF_WG)         \
    for (int i_ = 0; i_ < 8; i_++) g_step_prof[k][wg_][i_] = sp_[i_];
#if STEP_PROFILE >= 2
#define SPROF_W(k, slot)             \
  {                                  \
    __builtin_amdgcn_s_waitcnt(0);   \
    SPROF(k, slot)                   \
  }
#else
#define SPROF_W(k, slot)
#endif
extern "C" int hamsm_debug_step_profile(unsigned long long *h_out) {
  return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_step_prof), sizeof(unsigned long long) * 2 * SPROF_WG * 8) == hipSuccess ? 0 : -1;
}
#elif defined(STEP_TRACE)
// STEP_TRACE: the wall-clock time at which every launch of the two training kernels started (first workgroup), in launch
// order per kernel: shows bubbles in the launch stream without a profiler attached (tools/steptrace.py)
#define STRACE_N 32768
__device__ unsigned long long g_step_trace[2][STRACE_N];
__device__ unsigned g_step_trace_n[2];
#define SPROF_ID
#define SPROF(k, slot)                                                           \
  if (slot == 0 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {     \
    const unsigned i_ = atomicAdd(&g_step_trace_n[k], 1u);                       \
    if (i_ < STRACE_N) g_step_trace[k][i_] = wall_clock64();                     \
  }
#define SPROF_W(k, slot)
#define SPROF_FLUSH(k)
extern "C" int hamsm_debug_step_trace(unsigned long long *h_out, unsigned *h_n, int reset) {
  if (h_out && hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_step_trace), sizeof(unsigned long long) * 2 * STRACE_N) != hipSuccess) return -1;
  if (h_n && hipMemcpyFromSymbol(h_n, HIP_SYMBOL(g_step_trace_n), sizeof(unsigned) * 2) != hipSuccess) return -1;
  if (reset) {
    const unsigned z[2] = {0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_step_trace_n), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#else
#define SPROF_ID
#define SPROF(k, slot)
#define SPROF_W(k, slot)
#define SPROF_FLUSH(k)
#endif


struct StepG {  // a StepDev with its pointers in the global address space
  const GAS char *X;
  const GAS int32_t *rows;
  const GAS double *wf;
  long long n_b;
  GAS int32_t *labels;
  GAS double *Xg;
  GAS unsigned long long *mbits;
  GAS double *C, *counts, *csq, *Bpack;
  int K, flags, idx;
  GAS int32_t *report;
  __device__ __forceinline__ explicit StepG(const StepDev &d)
      : X(gp((const char *)d.X)), rows(gp(d.rows)), wf(gp(d.wf)), n_b(d.n_b), labels(gp(d.labels)), Xg(gp(d.Xg)),
        mbits(gp(d.mbits)), C(gp(d.C)), counts(gp(d.counts)), csq(gp(d.csq)), Bpack(gp(d.Bpack)), K(d.K),
        flags(d.flags), idx(d.idx), report(gp(d.report)) {}
};

template <bool BYVAL>
__device__ __forceinline__ StepDev load_step(const StepDev *steps, const StepPack &pk, unsigned i) {
  if constexpr (BYVAL) return pk.s[i];  // scalar loads from the kernel-argument segment
  static_assert(sizeof(StepDev) % 8 == 0, "StepDev is copied as 64-bit words");
  union {
    StepDev d;
    unsigned long long w[sizeof(StepDev) / 8];
  } u;
  const GAS unsigned long long *src = (const GAS unsigned long long *)(steps + i);
#pragma unroll
  for (unsigned q = 0; q < sizeof(StepDev) / 8; q++) u.w[q] = src[q];
  return u.d;
}

__device__ __forceinline__ StepDev load_step_mem(const StepDev *steps, unsigned i) {
  union {
    StepDev d;
    unsigned long long w[sizeof(StepDev) / 8];
  } u;
  const GAS unsigned long long *src = (const GAS unsigned long long *)(steps + i);
#pragma unroll
  for (unsigned q = 0; q < sizeof(StepDev) / 8; q++) u.w[q] = src[q];
  return u.d;
}

// E-step: one workgroup per RT 16-row tiles.  The stratum's centroid tiles are split over the 4 waves and stay in
// registers while the wave sweeps the workgroup's row tiles, so a workgroup pulls the centroid operands through L2
// once per 16*RT rows (with RT = 1 a call of 10^4 rows moved 40 MB of operands and the kernel was L2-bound).  All
// operand loads of a wave are issued before its first MFMA; the 4 partial first-minima are combined through LDS.
// Wave 3 also touches the lines the same rows of the NEXT call will gather; workgroup 0 of each stratum evaluates
// the low-count test of the stratum's previous step (the counts are final since that step's M-step).
template <typename XT, int KS, int JB, int RT, bool BYVAL>
__global__ __launch_bounds__(256) void k_step_assign(const StepDev *__restrict__ steps, const StepPack pk, int64_t ldx, int D,
                                                     double ratio, unsigned long long *stamp) {
  SPROF_ID
  SPROF(0, 0)
  // sampled launches (stamp != null, every 64th): first wave in / last wave out on the 100 MHz clock = the kernel's own duration
  if (stamp && threadIdx.x == 0) atomicMin(&stamp[0], (unsigned long long)wall_clock64());
  constexpr int RW = 16 * RT;  // rows per workgroup
  const StepDev sdesc = load_step<BYVAL>(steps, pk, blockIdx.y);
  const StepG st(sdesc);
  const long long r0 = (long long)blockIdx.x * RW;
  if (r0 >= st.n_b) return;
  SPROF_W(0, 1)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const GAS XT *X = (const GAS XT *)st.X;
  // prefetch, part 1 (wave 3): the next call's row index, requested before anything else of this wave so that
  // the in-order return of this wave's loads never waits for it alone
  const int prow = lane % RW, pslot = lane / RW;
  constexpr int PSTRIDE = 64 / RW;
  const int ynx = blockIdx.y < PACK_NY ? blockIdx.y : 0;
  const bool pf_on = BYVAL && wv == 3 && (int)blockIdx.y < pk.n_next && r0 + prow < pk.nx[ynx].n_b;
  long long fn = 0;
  if (pf_on) {
    const GAS int32_t *nrows = gp(pk.nx[ynx].rows);
    fn = nrows ? (long long)nrows[r0 + prow] : r0 + prow;
  }
  long long f[RT];
  bool ok[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    const long long r = r0 + rt * 16 + (lane & 15);
    ok[rt] = r < st.n_b;
    const long long rc = ok[rt] ? r : st.n_b - 1;  // (rows past the batch repeat its last row; nothing of them is kept)
    f[rt] = st.rows ? (long long)st.rows[rc] : rc;
  }
  SPROF_W(0, 2)
  // the first group of centroid operands does not depend on the row index: requested before the gather waits for it
  const int njt = (st.K + 15) >> 4;
  const double inf = __builtin_huge_val();
  double b[JB][KS], cs[JB][4];
  auto load_group = [&](int t0) {
#pragma unroll
    for (int u = 0; u < JB; u++) {
      // unconditional loads at a clamped tile (tiles past the last one are never multiplied: see the MFMA loop)
      const int jt = t0 + 4 * u < njt ? t0 + 4 * u : njt - 1;
#pragma unroll
      for (int q = 0; q < 4; q++) cs[u][q] = st.csq[jt * 16 + (lane >> 4) + 4 * q];
#pragma unroll
      for (int ks = 0; ks < KS; ks++) b[u][ks] = st.Bpack[((size_t)jt * KS + ks) * 64 + lane];
    }
  };
#ifdef STEP_TOUCH
  // cooperative warm-up: the stratum's operand lines were rewritten by the previous M-step (on other XCDs) and every
  // workgroup walks them in the same order, so the misses serialise; here each wave of an XCD first touches its own
  // share of the lines (one load instruction, one line per lane)
  double tv = 0.0;
  {
    const int nl_b = njt * KS * 4, nl_c = (st.K + 15) >> 4;  // 128-byte lines of Bpack and csq
    const int wpx = (((int)gridDim.x + 7) / 8) * 4;            // waves of this stratum on one XCD (round-robin dispatch)
    const int line = ((int)blockIdx.x / 8) * 4 + wv + lane * wpx;
    if (line < nl_b) tv = st.Bpack[(size_t)line * 16];
    else if (line < nl_b + nl_c) tv = st.csq[(line - nl_b) * 16];
  }
#endif
  load_group(wv);
  double a[RT][KS];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      // unconditional: rows past n_b read frame 0 and are not stored; features past D meet zero centroid fragments
      const int k = ks * 4 + (lane >> 4);
      a[rt][ks] = (double)X[f[rt] * ldx + (k < D ? k : D - 1)];
    }
  SPROF_W(0, 3)
  __shared__ double sd[4][RW];
  __shared__ int sj[4][RW];
  if (blockIdx.x == 0 && (st.flags & 6) && st.idx < *gp(sdesc.halt)) {
    // (the stratum's chain is live.)  One pass over the counts: for the previous step of the stratum
    // `counts < ratio * max(counts)` when the host predicted that none would be (_kmeans.py:1645), else
    // `(self._counts == 0).any()` before this step (:2037).  Either finding stops the chain HERE: this step's
    // M-step and everything after it see idx >= *halt.
    const GAS int32_t *prev = gp(sdesc.prev_report);
    int z = 0;
    double mx = -inf;
    for (int j = tid; j < st.K; j += 256) {
      double cj = st.counts[j];
      z |= (cj == 0.0);
      mx = cj > mx ? cj : mx;
    }
    int any_low = 0;
    if (st.flags & 4) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        double o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
      }
      if (lane == 0) sd[0][wv] = mx;
      __syncthreads();
      mx = sd[0][0];
#pragma unroll
      for (int w = 1; w < 4; w++) mx = sd[0][w] > mx ? sd[0][w] : mx;
      const double thr = ratio * mx;
      int low = 0;
      for (int j = tid; j < st.K; j += 256) low += st.counts[j] < thr;
      if (low) __hip_atomic_fetch_add((GAS int32_t *)prev + 1, low, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      any_low = __syncthreads_or(low);
    }
    const int any_z = __syncthreads_or((st.flags & 2) ? z : 0);  // (also: sd is reused below)
    if (tid == 0) {
      if (any_low) {
        *gp(sdesc.halt) = st.idx;
        *gp(sdesc.reason) = HAMSM_STOP_LOW;
      } else if (any_z) {
        st.report[0] = 1;
        *gp(sdesc.halt) = st.idx;
        *gp(sdesc.reason) = HAMSM_STOP_ZERO;
      }
    }
  }
  // centroids are the MFMA's A operand (rows), the 16 batch rows its B operand (columns): lane l accumulates, for
  // batch row l&15, the centroids (l>>4) + 4q of each tile -- the first-minimum is register-local
  double bd[RT];
  int bi[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    bd[rt] = inf;
    bi[rt] = 0x7fffffff;
  }
  XT pfv[PF_MAX];
#pragma unroll
  for (int i = 0; i < PF_MAX; i++) pfv[i] = (XT)0;
  bool pf_todo = pf_on;
  for (int t0 = wv; t0 < njt; t0 += 4 * JB) {
    if (t0 != wv) load_group(t0);
    if (pf_todo) {
      // prefetch, part 2: one dword per 128-byte line of the next call's row, behind this wave's last operand loads
      pf_todo = false;
      const GAS XT *Xn = gp((const XT *)pk.nx[ynx].X) + fn * ldx;
      const int per = 128 / (int)sizeof(XT);
#pragma unroll
      for (int i = 0; i < PF_MAX; i++) {
        const int slot = pslot + PSTRIDE * i;
        int k = slot * per;
        k = k < D ? k : D - 1;  // the last element covers a row end that straddles into one more line
        pfv[i] = Xn[k];         // (unconditional: slots past the row touch its last element again)
      }
    }
    SPROF_W(0, 4)
#pragma unroll
    for (int u = 0; u < JB; u++) {
      int jt = t0 + 4 * u;
      if (jt < njt) {
        v4f64 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) acc[rt] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
          for (int rt = 0; rt < RT; rt++) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[u][ks], a[rt][ks], acc[rt], 0, 0, 0);
        const int j0 = jt * 16 + (lane >> 4);
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            double d = __builtin_fma(-2.0, acc[rt][q], cs[u][q]);
            bool lt = d < bd[rt];  // a lane's centroids ascend (q, then tile), so strict < keeps its lowest j
            bd[rt] = lt ? d : bd[rt];
            bi[rt] = lt ? j0 + 4 * q : bi[rt];
          }
      }
    }
  }
  // the batch rows as f64 for the M-step: row tile rt is written by wave (rt + 1) & 3
  // (no copy where the M-step fetches its member rows from the frames: HAMSM_STEP_DIRECT_PLAIN)
  if (st.Xg)
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
    if (((rt + 1) & 3) == wv) {
      // rows of 4 KS doubles (whole 128-byte lines), the batch padded to whole tiles: unconditional stores
      const long long r = r0 + rt * 16 + (lane & 15);
#pragma unroll
      for (int ks = 0; ks < KS; ks++) st.Xg[(size_t)r * (4 * KS) + ks * 4 + (lane >> 4)] = a[rt][ks];
    }
  SPROF_W(0, 5)
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    double d = bd[rt];
    int j = bi[rt];
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {  // lanes l, l^16, l^32, l^48 hold the same batch row
      double d2 = __shfl_xor(d, off, 64);
      int j2 = __shfl_xor(j, off, 64);
      bool take = (d2 < d) || (d2 == d && j2 < j);
      d = take ? d2 : d;
      j = take ? j2 : j;
    }
    if (lane < 16) {
      sd[wv][rt * 16 + lane] = d;
      sj[wv][rt * 16 + lane] = j;
    }
  }
  __syncthreads();
  SPROF_W(0, 6)
  if (tid < RW && r0 + tid < st.n_b) {
    double d = sd[0][tid];
    int j = sj[0][tid];
#pragma unroll
    for (int w = 1; w < 4; w++) {
      double d2 = sd[w][tid];
      int j2 = sj[w][tid];
      bool take = (d2 < d) || (d2 == d && j2 < j);
      d = take ? d2 : d;
      j = take ? j2 : j;
    }
    if (j == 0x7fffffff) j = 0;  // all-NaN row: label 0 like the oracle
    st.labels[r0 + tid] = j;
    const long long r = r0 + tid;
    // (result unused: a non-returning atomic -- the workgroup does not wait for it)
    __hip_atomic_fetch_or(&st.mbits[(size_t)j * sdesc.wstride + (r >> 6)], 1ull << (r & 63), __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT);
  }
  if (pf_on) {
#pragma unroll
    for (int i = 0; i < PF_MAX; i++) asm volatile("" ::"v"(pfv[i]));  // the touched lines are only wanted in L2
  }
#ifdef STEP_TOUCH
  asm volatile("" ::"v"(tv));
#endif
  SPROF_W(0, 7)
#if defined(STEP_PROFILE) && STEP_PROFILE < 2
  SPROF(0, 7)
#endif
  SPROF_FLUSH(0)
  if (stamp && threadIdx.x == 0) atomicMax(&stamp[1], (unsigned long long)wall_clock64());
}

// members m0 .. m0+W-1 of a sorted member list held in the wave's LDS strip (srow[q] = batch row, swl[q] = weight of
// member q; broadcast reads -- v_readlane into an address or an operand costs ~100 cycles each here): all W rows are
// requested before the first add; branch-free -- past the end the last member is fetched again (same lines) and its
// add is deselected
template <int W, int H>
__device__ __forceinline__ void upd_add_listed(double (&acc)[H], const GAS double *Xg, int ldg, const unsigned *srow,
                                               const double *swl, bool weighted, int m0, int mc, int D, int lane) {
  double xv[W][H];
#pragma unroll
  for (int q = 0; q < W; q++) {
    const int mq = m0 + q < mc ? m0 + q : mc - 1;
    const size_t rq = (size_t)srow[mq];
#pragma unroll
    for (int h = 0; h < H; h++) {
      const int k = lane + 64 * h;
      xv[q][h] = Xg[rq * ldg + (k < D ? k : D - 1)];  // unconditional; lanes past D carry a copy that is never stored
    }
  }
#pragma unroll
  for (int q = 0; q < W; q++) {
    const int mq = m0 + q < mc ? m0 + q : mc - 1;
    const double wq = weighted ? swl[mq] : 1.0;
#pragma unroll
    for (int h = 0; h < H; h++) {
      double nv = acc[h] + xv[q][h] * wq;
      acc[h] = (m0 + q < mc) ? nv : acc[h];
    }
  }
}

// the same from the frames themselves: srow holds FRAME indices (batch rows translated through the call's row list), a row is
// `ldx` values of XT, widened in registers -- no copy of the batch is written by the E-step and read back (round 5)
template <int W, int H, typename XT>
__device__ __forceinline__ void upd_add_direct(double (&acc)[H], const GAS XT *X, int64_t ldx, const unsigned *srow, const double *swl,
                                               bool weighted, int m0, int mc, int D, int lane) {
  XT xv[W][H];
#pragma unroll
  for (int q = 0; q < W; q++) {
    const int mq = m0 + q < mc ? m0 + q : mc - 1;
    const size_t rq = (size_t)srow[mq];
#pragma unroll
    for (int h = 0; h < H; h++) {
      const int k = lane + 64 * h;
      xv[q][h] = X[rq * ldx + (k < D ? k : D - 1)];  // unconditional; lanes past D carry a copy that is never stored
    }
  }
#pragma unroll
  for (int q = 0; q < W; q++) {
    const int mq = m0 + q < mc ? m0 + q : mc - 1;
    const double wq = weighted ? swl[mq] : 1.0;
#pragma unroll
    for (int h = 0; h < H; h++) {
      double nv = acc[h] + (double)xv[q][h] * wq;
      acc[h] = (m0 + q < mc) ? nv : acc[h];
    }
  }
}

// ---- the E-step of MFMA-bound calls: centroids RANKED on the bf16 matrix unit, f64 only where the ranking cannot decide ------------
// At BASELINE config 5's shape (12 500-row batches x 1250 centroids x 64 features per stratum and call) k_step_assign is bound by
// the f64 matrix unit (0.47 of its peak, 107 us per launch of two strata) and the training pass is 80 % of the step.  A label is an
// argmin: it needs the ORDER of the distances.  So, as in hamsm_assign's ranking pass (assign.hip), the scores
//     s_j = x . c_j - csq_j / 2 = -d_j / 2
// are formed on v_mfma_f32_32x32x16_bf16 from operands that are exact sums of bf16 pieces (truncation splits, 8 significant bits
// each; the products x1 c1 + x1 c2 + x2 c1 of the light form: 3 MFMAs per 16 feature slots), every |(-2 s_j) - d_j| is below a
// PROVEN bound E, and a row's label is final only if its two best scores are further apart than 2 E (x 1.01).  What is new here:
//   * the fragments are MAINTAINED: the run packs them once (k_rank_pack), then the M-step wave that rewrites centroid j rewrites
//     its fragment slots (rank_store_row), and so does the device re-seeding -- no conversion pass, no gather on the E-step's path
//     (that is what sank the composition of DESIGN.md 5.3c and the on-the-fly conversion of round 1);
//   * the ranking keeps the three best scores and the indices of the best two (4 vector instructions per ranked value).  If the
//     third-best is out of reach of the best (2 (M1 - M3) > 2 E) but the second is not, the exact argmin is one of TWO known
//     centroids: the row's thread evaluates the contract's two f64 fma chains itself and takes the first minimum -- 2 x D fmas
//     instead of a sweep over all K centroids.  Only a row whose three best are within reach of each other, or whose data are not
//     finite, is swept exactly against every centroid by the workgroup (rare: counted in `rk.stat`);
//   * the bound's scale max_j ||c_j||^2 cannot be read off a pack kernel (the centroids move every step): a centroid is always a
//     convex combination of the centroids the run started with and of batch rows it has seen (_k_means_minibatch.pyx:80-104 with
//     weights >= 0; re-seeding copies a batch row), so max(||c||^2 at the start, ||x||^2 of every row seen so far) bounds it --
//     kept as f32 bit patterns in RANK_SLOTS words per stratum (atomic max by the E-steps, which compute ||x||^2 anyway).
// Arithmetic of the exact parts: d = fma(-2, chain, csq), chain = fma over ascending k from +0.0 -- the contract (hamsm.h), what
// v_mfma_f64_16x16x4_f64 computes in k_step_assign; first strict minimum over ascending j.  Labels are therefore those of
// k_step_assign bit for bit; tests/test_ranked_training.py compares whole passes with the oracle with this path forced.
// Layout: a workgroup = 128 batch rows (four tiles of 32) x 8 waves.  Prologue: waves w and w + 4 turn row tile w into operand pieces
// (from global memory straight into registers, as k_assign_filter's bf16 forms) and leave them in LDS as the B-operand images all
// eight waves read.  Sweep: wave w takes the centroid tiles w, w + 8, ... -- fragments (A operand) from L2 into registers, one tile
// ahead, -csq / 2 as the accumulator's initial value -- against all four row tiles: 12 KB MFMAs per (tile, row tile), then the 64
// ranking instructions; two waves per SIMD, so one's ranking overlaps the other's MFMAs.  No barrier inside the sweep; the eight
// partial top-3 per row are merged through LDS (the operand images' bytes, re-used).
// What was measured on the way (C5-shaped pass, profiles/r04_*, tools/rankprof.py: phase stamps per workgroup):
//   waves split the tiles, 64 rows per workgroup, rows' pieces in every wave's registers: E-step 87 -> 60 us per launch (107 us for
//     k_step_assign); the exact parts were the tail (per-row sweeps over all centroids by the workgroup: 10 us each, one workgroup
//     130 us) until rows between two candidates went to one wave per row and the rest to the f64 matrix unit, 16 rows at a time;
//   waves own row tiles, fragments through an LDS ring by global_load_lds (k_assign_filter's shape): 72 us -- one barrier per group
//     of 3 tiles, and a group's sweep (0.65 us) is shorter than the L2 -> LDS latency;
//   this layout with 4 waves: 60 us, sweep 23.5 us per workgroup of which ranking 11 (one wave per SIMD: its MFMAs and its ranking
//     do not overlap; interleaving them in program order between scheduling fences: 32 us); with 8 waves: sweep 14.5 us, 45 us per
//     launch.  What is left of the sweep is the fragment stream itself: ~9 us with the MFMAs compiled out (62 MB per launch out
//     of L2, every workgroup the same 320 KB per stratum).
#define RANK_SLOTS 64
#ifdef RANK_PROFILE  // (diagnostic build only, tools/rankprof.py: where a workgroup of the ranked E-step spends its time, 100 MHz ticks)
__device__ unsigned long long g_rank_prof[16];
#define RPROF_INIT unsigned long long rp_t_ = wall_clock64(), rp_n_;
#define RPROF(i)                                                        \
  {                                                                     \
    rp_n_ = wall_clock64();                                             \
    if (threadIdx.x == 0) atomicAdd(&g_rank_prof[i], rp_n_ - rp_t_);    \
    if (threadIdx.x == 0 && (i) == 4) atomicMax(&g_rank_prof[9], rp_n_ - rp_t_); \
    rp_t_ = rp_n_;                                                      \
  }
#define RPROF_END if (threadIdx.x == 0) atomicAdd(&g_rank_prof[8], 1ull);
extern "C" int hamsm_debug_rank_profile(unsigned long long *h_out, int reset) {
  if (h_out && hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_rank_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_rank_prof), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#else
#define RPROF_INIT
#define RPROF(i)
#define RPROF_END
#endif
struct RankPack {
  unsigned *apk[PACK_NY];        // [tile32][KB][2 pieces][64 lanes][4 words]: lane l = centroid l % 32, slots 16 m + 8 (l / 32) + i
  float *hcs[PACK_NY];           // [tiles32 x 32] -csq_j / 2 as f32: the scores' initial accumulator (-3e38 past K: never ranks first)
  unsigned *bnd[PACK_NY];        // [RANK_SLOTS] f32 bit patterns, max = upper bound of every ||c_j||^2 (and |csq_j|) of the stratum
  unsigned long long *stat;      // [4] rows certified / settled between two candidates / swept exactly / (unused)   (nullable)
};
typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf16 __attribute__((ext_vector_type(8)));
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

// fragment slots of ONE centroid row: lane k < D holds feature k's new value; `csq` (all lanes) its new ||c||^2
__device__ __forceinline__ void rank_store_row(GAS unsigned short *apk, GAS float *hcs, int KB, int j, int lane, int D, double vk,
                                               double csq) {
  const size_t tile = (size_t)(j >> 5) * KB;
  const int row = j & 31;
  auto put = [&](int k, float vf) {
    const unsigned b0 = __float_as_uint(vf), p1 = b0 & 0xffff0000u;
    const float r1 = vf - __uint_as_float(p1);
    const size_t at = (((tile + (k >> 4)) * 2) * 64 + (row + 32 * ((k >> 3) & 1))) * 8 + (k & 7);  // piece 0; piece 1 is 64 * 8 further
    apk[at] = (unsigned short)(p1 >> 16);
    apk[at + 512] = (unsigned short)(__float_as_uint(r1) >> 16);
  };
  if (lane < D) put(lane, (float)vk);
  if (lane == 0) hcs[j] = (float)(-0.5 * csq);
}

// the run's fragments and bounds from the strata's tables (one wave per 32-centroid tile)
__global__ __launch_bounds__(64) void k_rank_pack(const double *__restrict__ C, const double *__restrict__ csq, int K, int D, int KB,
                                                  unsigned *__restrict__ apk, float *__restrict__ hcs, unsigned *__restrict__ bnd) {
  const int jt = blockIdx.x, lane = threadIdx.x, row = lane & 31, h = lane >> 5, j = jt * 32 + row;
  const bool live = j < K;
  const double cs = live ? csq[j] : 0.0;
  double n2 = 0.0;
  for (int m = 0; m < KB; m++) {
    unsigned w[2][4];
    for (int i = 0; i < 8; i++) {
      const int k = 16 * m + 8 * h + i;
      const double v = (live && k < D) ? C[(size_t)j * D + k] : 0.0;
      n2 += v * v;
      const float vf = (float)v;
      const unsigned p1 = __float_as_uint(vf) & 0xffff0000u;
      const unsigned p2 = __float_as_uint(vf - __uint_as_float(p1)) & 0xffff0000u;
      if (i & 1) {
        w[0][i >> 1] |= p1;
        w[1][i >> 1] |= p2;
      } else {
        w[0][i >> 1] = p1 >> 16;
        w[1][i >> 1] = p2 >> 16;
      }
    }
    for (int q = 0; q < 2; q++)
      for (int z = 0; z < 4; z++) apk[((((size_t)jt * KB + m) * 2 + q) * 64 + lane) * 4 + z] = w[q][z];
  }
  n2 += __shfl_xor(n2, 32, 64);
  // (a row past K never ranks first; finite: the index bits in the ranked values must not turn an infinity into a signalling NaN)
  if (h == 0) hcs[j] = live ? (float)(-0.5 * cs) : -3.0e38f;
  if (h == 0 && live) {  // (non-negative floats order like their bit patterns; NaN patterns sort above +inf and poison the bound)
    const float a = (float)n2 * 1.000001f, b = __builtin_fabsf((float)cs) * 1.000001f;
    atomicMax(&bnd[j & (RANK_SLOTS - 1)], __float_as_uint(a > b || !(b == b) ? a : b));
  }
}

// (A1 >= A2 >= A3 with the indices of A1, A2) merged with another such triple: the three largest of the six, indices of the two
// largest; equal values: the lower index first
__device__ __forceinline__ void rank_merge3(float &A1, float &A2, float &A3, int &ja1, int &ja2, float B1, float B2, float B3, int jb1,
                                            int jb2) {
  if (B1 > A1 || (B1 == A1 && jb1 < ja1)) {
    float t;
    int u;
    t = A1, A1 = B1, B1 = t;
    t = A2, A2 = B2, B2 = t;
    t = A3, A3 = B3, B3 = t;
    u = ja1, ja1 = jb1, jb1 = u;
    u = ja2, ja2 = jb2, jb2 = u;
  }
  if (B1 > A2 || (B1 == A2 && jb1 < ja2)) {
    A3 = __builtin_fmaxf(A2, B2);
    A2 = B1;
    ja2 = jb1;
  } else
    A3 = __builtin_fmaxf(A3, B1);
}

#define RANK_NW 8  // waves per workgroup of the ranked E-step: two per SIMD, so that one's ranking (vector unit) overlaps the other's MFMAs
// NRT = row tiles of 32 batch rows per workgroup.  4 (128 rows): round 4's shape.  8 (256 rows, round 5; runs without the f64 batch
// copy only): at C5's shape a launch of two strata is 98 workgroups instead of 196, so the E-steps of BOTH queues (all four strata,
// 196 workgroups) are resident on the 256 CUs at once -- with 128-row workgroups the two queues' launches (392 workgroups, one per
// CU: 226 registers x 8 waves) took turns, two rounds of a ~28 us workgroup per call -- and every workgroup streams its stratum's
// 320 KB of fragments through L2 once per 256 rows instead of once per 128.
template <typename XT, int KB, int NRT, int NWV = RANK_NW, bool PIPE = false>
__global__ __launch_bounds__(NWV * 64, 2) void k_step_assign_ranked(const StepPack pk, const RankPack rk, int64_t ldx, int D, int KS,
                                                                        double ratio, float c1, float c2, unsigned long long *stamp) {
  constexpr int NW = NWV, NT = NW * 64;
  constexpr int RW = 32 * NRT;  // rows per workgroup
  static_assert((2 * NRT) % NW == 0 && NRT <= NW * 2 && (NRT == 4 || NRT == 8), "prologue: 2 NRT (row tile, block parity) items over the waves");
  constexpr int NIT = 2 * NRT / NW;  // prologue items per wave
  const StepDev sdesc = pk.s[blockIdx.y];
  const StepG st(sdesc);
  const long long r0 = (long long)blockIdx.x * RW;
  if (r0 >= st.n_b) return;
  // A stratum whose chain has stopped (an earlier launch set its stop word: visible here by stream order) only drains: this step
  // will not be applied, nobody reads its labels.  At this kernel's size that matters -- the fused first steps stop nearly every
  // stratum of a C5-shaped pass at once, and a 1000-call run then drained at the full 45 us per launch (the f64 kernel of the
  // small shapes keeps working through: 10 us launches, rare stops, and the extra dependent load would sit on its critical path).
  if (st.idx >= *gp(sdesc.halt)) return;
  if (stamp && threadIdx.x == 0) atomicMin(&stamp[0], (unsigned long long)wall_clock64());
  RPROF_INIT
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, row = lane & 31, h = lane >> 5;
  const GAS XT *X = (const GAS XT *)st.X;
  const GAS v4u32 *apk = (const GAS v4u32 *)gp(rk.apk[blockIdx.y]);
  const GAS float *hcs = gp(rk.hcs[blockIdx.y]);
  GAS unsigned *bndw = gp(rk.bnd[blockIdx.y]);
  const float ninf = -__builtin_huge_valf();
  const double inf = __builtin_huge_val();
  // LDS: the four row tiles' operand images (B operand, as the lanes hold them) during the sweep; afterwards the same bytes hold
  // the waves' partial top-3 and the strips of the two-candidate chains
  constexpr int SB_BYTES = NRT * KB * 2 * 64 * 16, SM_BYTES = NW * RW * 3 * 4, SJ_BYTES = NW * RW * 2 * 4, ST_BYTES = NW * 3 * 64 * 8;
  constexpr int BUF_BYTES = SB_BYTES > SM_BYTES + SJ_BYTES + ST_BYTES ? SB_BYTES : SM_BYTES + SJ_BYTES + ST_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char s_buf[BUF_BYTES];
  v4u32(*s_b)[KB][2][64] = (v4u32(*)[KB][2][64])s_buf;                    // [NRT][KB][2][64]
  float(*s_m)[RW][3] = (float(*)[RW][3])s_buf;                            // [NW][RW][3]
  int(*s_j)[RW][2] = (int(*)[RW][2])(s_buf + SM_BYTES);                   // [NW][RW][2]
  double(*s_t)[3][64] = (double(*)[3][64])(s_buf + SM_BYTES + SJ_BYTES);  // [NW][3][64]
  __shared__ float s_x[2][RW];
  __shared__ int s_full[RW], s_nfull, s_two[RW][3], s_ntwo, s_lab[RW];
  __shared__ double s_d[NW];
  __shared__ double s_fd[NW][16];
  __shared__ int s_fi[NW][16];
  if (tid == 0) s_nfull = s_ntwo = 0;
  if (tid < RW) s_lab[tid] = -1;
  const int njt = (st.K + 31) >> 5;
  // ---- prologue: waves w and w + 4 turn row tile w into operand pieces, the 16-slot blocks of even / odd number each (lane l:
  // batch row l % 32 of the tile, the 8 feature slots 16 m + 8 (l / 32) + i of a block -- from global memory straight into
  // registers, as k_assign_filter's bf16 forms) and leave them in LDS for all eight waves ----
  // (item `it` of wave wv: row tile (wv + NW it) % NRT, block parity (wv + NW it) / NRT -- NRT = 4: one item per wave, waves w and
  // w + 4 share row tile w; NRT = 8: wave w converts both parities of row tile w)
  const int rtw = wv % NRT;
  const long long rmine = r0 + 32 * rtw + row;
  const bool ok = rmine < st.n_b;
  const long long rcl = ok ? rmine : st.n_b - 1;  // (rows past the batch repeat its last row; nothing of them is kept)
  const long long fr = st.rows ? (long long)st.rows[rcl] : rcl;
  // (the row of the stratum's NEXT call that this lane will touch towards this XCD's L2, as k_step_assign's wave 3 does)
  const bool pf_on = wv < NRT && (int)blockIdx.y < pk.n_next && rmine < pk.nx[blockIdx.y].n_b;
  long long fn = 0;
  if (pf_on) {
    const GAS int32_t *nrows = gp(pk.nx[blockIdx.y].rows);
    fn = nrows ? (long long)nrows[rmine] : rmine;
  }
  unsigned bu = bndw[lane & (RANK_SLOTS - 1)];
  // this wave's first centroid tile does not depend on the rows: requested before the gather waits for the row indices.
  // Tiles are walked from a start that differs from workgroup to workgroup (every workgroup of a stratum reads the same 256 B
  // per centroid: walking them in the same order at the same time queues them all on the same L2 channels)
  const int nmy = (njt - wv + NW - 1) / NW;  // tiles of this wave: wv, wv + NW, ...  (<= 0: none)
  const int rot = nmy > 0 ? (int)(blockIdx.x % (unsigned)nmy) : 0;
  struct Frag {
    v4u32 c[KB][2];
    v4f32 i0[4];  // the accumulator's initial value: -csq / 2 of the lane's 16 centroids
  };
  auto tile_of = [&](int i) {  // i-th tile of this wave's walk
    int q = i + rot;
    q = q >= nmy ? q - nmy : q;
    return wv + NW * q;
  };
  auto load_frag = [&](int i) {
    Frag f;
    const int t = nmy > 0 ? tile_of(i < nmy ? i : nmy - 1) : 0;
    const int tc = t < njt ? t : njt - 1;
    const GAS v4u32 *ap = apk + (size_t)tc * (KB * 2 * 64) + lane;
#pragma unroll
    for (int m = 0; m < KB; m++) {
      f.c[m][0] = ap[(m * 2 + 0) * 64];
      f.c[m][1] = ap[(m * 2 + 1) * 64];
    }
    const GAS v4f32 *hp = (const GAS v4f32 *)(hcs + (size_t)tc * 32 + 4 * h);
#pragma unroll
    for (int g = 0; g < 4; g++) f.i0[g] = hp[2 * g];  // centroids 8 g + 4 h + (0..3) of the tile
    return f;
  };
  Frag cur = load_frag(0);
  typedef XT xt4u __attribute__((ext_vector_type(4), aligned(sizeof(XT))));
  constexpr int KBH = (KB + 1) / 2;  // blocks per item
  XT raw[NIT][KBH][8];               // (NRT = 4: kept to the end, the batch copy for the M-step is written from them)
  {
    const GAS XT *rp = X + fr * ldx;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int mpar = (wv + NW * it) / NRT;  // (wave-uniform)
#pragma unroll
      for (int mm = 0; mm < KBH; mm++) {
        const int m = 2 * mm + mpar;
        const int k0 = 16 * m + 8 * h;
        if (m < KB && 16 * m + 16 <= D) {  // both half-waves stay inside the row
          const xt4u va = *(const GAS xt4u *)(rp + k0), vb = *(const GAS xt4u *)(rp + k0 + 4);
#pragma unroll
          for (int i = 0; i < 4; i++) {
            raw[it][mm][i] = va[i];
            raw[it][mm][4 + i] = vb[i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; i++) {  // (slots past the row arrive as zeros; the load itself stays inside the row)
            const int k = k0 + i;
            const XT v = rp[k < D ? k : D - 1];
            raw[it][mm][i] = (m < KB && k < D) ? v : (XT)0;
          }
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int mpar = (wv + NW * it) / NRT;
      float xn2 = 0.f;
#pragma unroll
      for (int mm = 0; mm < KBH; mm++) {
        const int m = 2 * mm + mpar;
        v4u32 p1, p2;
#pragma unroll
        for (int w = 0; w < 4; w++) {
          float a0[2], a1[2];
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const float v = (float)raw[it][mm][2 * w + i];
            xn2 = __builtin_fmaf(v, v, xn2);
            a0[i] = v;
            a1[i] = v - __uint_as_float(__float_as_uint(v) & 0xffff0000u);
          }
          // pieces by truncation: piece 1 = the upper half of the f32 pattern, piece 2 = the upper half of the exact remainder; two
          // slots share a word, picked by ONE v_perm_b32.  A value that is not finite leaves a NaN in piece 2 (inf - inf, NaN - x)
          p1[w] = __builtin_amdgcn_perm(__float_as_uint(a0[1]), __float_as_uint(a0[0]), 0x07060302u);
          p2[w] = __builtin_amdgcn_perm(__float_as_uint(a1[1]), __float_as_uint(a1[0]), 0x07060302u);
        }
        if (m < KB) {
          s_b[rtw][m][0][lane] = p1;
          s_b[rtw][m][1][lane] = p2;
        }
      }
      xn2 += __shfl_xor(xn2, 32, 64);
      if (lane < 32) s_x[mpar][32 * rtw + row] = xn2;  // (the row's ||x||^2 = the two parities' parts, summed by whoever needs it)
    }
  }
  if (blockIdx.x == 0 && (st.flags & 6) && st.idx < *gp(sdesc.halt)) {
    // (k_step_assign's block: the stratum's chain is live.)  `counts < ratio * max(counts)` for the stratum's previous
    // step when the host predicted that none would be low (_kmeans.py:1645), `(counts == 0).any()` before this one (:2037)
    const GAS int32_t *prev = gp(sdesc.prev_report);
    int z = 0;
    double mx = -inf;
    for (int j = tid; j < st.K; j += NT) {
      double cj = st.counts[j];
      z |= (cj == 0.0);
      mx = cj > mx ? cj : mx;
    }
    int any_low = 0;
    if (st.flags & 4) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        double o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
      }
      if (lane == 0) s_d[wv] = mx;
      __syncthreads();
      mx = s_d[0];
#pragma unroll
      for (int w = 1; w < NW; w++) mx = s_d[w] > mx ? s_d[w] : mx;
      const double thr = ratio * mx;
      int low = 0;
      for (int j = tid; j < st.K; j += NT) low += st.counts[j] < thr;
      if (low) __hip_atomic_fetch_add((GAS int32_t *)prev + 1, low, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      any_low = __syncthreads_or(low);
    }
    const int any_z = __syncthreads_or((st.flags & 2) ? z : 0);
    if (tid == 0) {
      if (any_low) {
        *gp(sdesc.halt) = st.idx;
        *gp(sdesc.reason) = HAMSM_STOP_LOW;
      } else if (any_z) {
        st.report[0] = 1;
        *gp(sdesc.halt) = st.idx;
        *gp(sdesc.reason) = HAMSM_STOP_ZERO;
      }
    }
  }
  __syncthreads();  // the row tiles' operand images are in LDS
  RPROF(0)
  XT pfv[2] = {(XT)0, (XT)0};
  if (pf_on) {  // (one dword per 128-byte line of the next call's row; lanes l and l + 32 share a row: every other line each)
    const GAS XT *Xn = gp((const XT *)pk.nx[blockIdx.y].X) + fn * ldx;
    const int per = 128 / (int)sizeof(XT);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      int k = (h + 2 * i) * per;
      k = k < D ? k : D - 1;
      pfv[i] = Xn[k];
    }
  }
  // ---- the sweep: this wave's centroid tiles against ALL four row tiles (12 KB MFMAs per tile of fragments: the next tile's
  // loads have that long to arrive); per row tile the three largest scores per row (and half-wave), the tiles of the largest
  // two.  The MFMAs of a (tile, row tile) unit are followed by its 64 ranking instructions: within one wave they do not overlap,
  // the SIMD's other wave fills the matrix unit meanwhile (measured: interleaving them in one wave's instruction order between
  // scheduling fences is slower -- the fences also pin the operand reads from LDS in front of their MFMAs) ----
  float M1[NRT], M2[NRT], M3[NRT];
  int J1[NRT], J2[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; rt++) {
    M1[rt] = M2[rt] = M3[rt] = ninf;
    J1[rt] = J2[rt] = 0;
  }
  // one accumulator register into the running top-3 of its row tile: the register's number rides in the 4 low mantissa bits (the
  // bound charges the 16 ulp), so the maxima know their own index.  Written as instructions: through the builtins the compiler
  // canonicalises v first.
  auto rank1 = [&](float &m1, float &m2, float &m3, float a, int r) {
    const float v = __uint_as_float((__float_as_uint(a) & 0xfffffff0u) | (unsigned)r);
    float m3n, m2n, m1n;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(m3n) : "v"(m2), "v"(m3), "v"(v));
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(m2n) : "v"(m1), "v"(m2), "v"(v));
    asm("v_max_f32 %0, %1, %2" : "=v"(m1n) : "v"(m1), "v"(v));
    m3 = m3n;
    m2 = m2n;
    m1 = m1n;
  };
  // whose tile?  (a value whose bits equal the old maximum's while a larger one arrived in this tile may be the demoted old maximum
  // or a twin of it from this tile: then both are among the three best, M2 == M3, and the row is never settled between two
  // candidates.)  Selects, not branches: the compiled form of the conditional expressions was three exec-mask regions per unit.
  auto whose = [&](int rt_, unsigned o1, unsigned o2, int t) {
    const unsigned n1 = __float_as_uint(M1[rt_]), n2 = __float_as_uint(M2[rt_]);
    const int c1 = -(int)(n1 != o1), c2 = -(int)(n2 != o2), dm = -(int)(n2 == o1);
    const int j2n = (J1[rt_] & (c1 & dm)) | (t & ~(c1 & dm));
    J2[rt_] = (j2n & c2) | (J2[rt_] & ~c2);
    J1[rt_] = (t & c1) | (J1[rt_] & ~c1);
  };
  if constexpr (PIPE) {
    // Round 5: the sweep as a software pipeline.  The compiled form of the plain loop below is strictly phased per (tile, row tile)
    // unit -- 8 x (ds_read_b128, s_waitcnt, MFMA...) with every LDS latency exposed, then the 64 ranking instructions, then three
    // branch regions -- ~1 780 cycles per unit against 384 of MFMA time, and the SIMD's two waves run the same phases at the same
    // time.  Here the MFMAs of unit u + 1 are issued BETWEEN the ranking instructions of unit u (one MFMA, then ~5 ranking
    // instructions: the wave issues in order, so only fine interleaving lets its own vector work run under its own matrix work),
    // its operand images are read from LDS one 16-slot block ahead, and the order is pinned by scheduling fences.
    auto ldb = [&](int rt_, int m, v4u32 &b1, v4u32 &b2) {
      b1 = s_b[rt_][m][0][lane];
      b2 = s_b[rt_][m][1][lane];
    };
    auto init_of = [&](const Frag &f) {
      v16f32 z;
#pragma unroll
      for (int r = 0; r < 16; r++) z[r] = f.i0[r >> 2][r & 3];
      return z;
    };
    Frag nxt = load_frag(1);
    v16f32 acc_cur = init_of(cur);
    if (nmy > 0) {  // unit (0, 0) has nothing to hide behind
#pragma unroll
      for (int m = 0; m < KB; m++) {
        v4u32 b1, b2;
        ldb(0, m, b1, b2);
        const v8bf16 a1 = __builtin_bit_cast(v8bf16, cur.c[m][0]), a2 = __builtin_bit_cast(v8bf16, cur.c[m][1]);
        acc_cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(v8bf16, b2), acc_cur, 0, 0, 0);
        acc_cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, __builtin_bit_cast(v8bf16, b1), acc_cur, 0, 0, 0);
        acc_cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(v8bf16, b1), acc_cur, 0, 0, 0);
      }
    }
    for (int i = 0; i < nmy; i++) {
      const int t = tile_of(i);
#pragma unroll
      for (int rt = 0; rt < NRT; rt++) {
        constexpr int NM = 3 * KB;  // MFMAs per unit
        const bool last_rt = rt == NRT - 1;
        const bool have_next = !last_rt || i + 1 < nmy;  // (wave-uniform)
        const int rtn = last_rt ? 0 : rt + 1;
        const unsigned o1 = __float_as_uint(M1[rt]), o2 = __float_as_uint(M2[rt]);
        v16f32 acc_nxt;
        if (have_next) {
          const Frag &fa = last_rt ? nxt : cur;
          acc_nxt = init_of(fa);
          v4u32 b1, b2, nb1, nb2;
          ldb(rtn, 0, b1, b2);
#pragma unroll
          for (int m = 0; m < KB; m++) {
            if (m + 1 < KB) ldb(rtn, m + 1, nb1, nb2);
            const v8bf16 a1 = __builtin_bit_cast(v8bf16, fa.c[m][0]), a2 = __builtin_bit_cast(v8bf16, fa.c[m][1]);
#pragma unroll
            for (int q = 0; q < 3; q++) {
              const int k = 3 * m + q;
              if (q == 0) acc_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(v8bf16, b2), acc_nxt, 0, 0, 0);
              if (q == 1) acc_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, __builtin_bit_cast(v8bf16, b1), acc_nxt, 0, 0, 0);
              if (q == 2) acc_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(v8bf16, b1), acc_nxt, 0, 0, 0);
#pragma unroll
              for (int r = 16 * k / NM; r < 16 * (k + 1) / NM; r++) rank1(M1[rt], M2[rt], M3[rt], acc_cur[r], r);
              __builtin_amdgcn_sched_barrier(0);
            }
            b1 = nb1;
            b2 = nb2;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) rank1(M1[rt], M2[rt], M3[rt], acc_cur[r], r);
          acc_nxt = acc_cur;
        }
        whose(rt, o1, o2, t);
        acc_cur = acc_nxt;
      }
      cur = nxt;
      nxt = load_frag(i + 2);  // (clamped; unused past the last tile)
    }
  } else {
  Frag nxt_unused = cur;
  (void)nxt_unused;
  for (int i = 0; i < nmy; i++) {
    const int t = tile_of(i);
    const Frag nxt = load_frag(i + 1);  // (clamped; unused past the last tile)
    v16f32 init;
#pragma unroll
    for (int r = 0; r < 16; r++) init[r] = cur.i0[r >> 2][r & 3];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
      v16f32 acc;
#pragma unroll
      for (int m = 0; m < KB; m++) {
        const v8bf16 a1 = __builtin_bit_cast(v8bf16, cur.c[m][0]), a2 = __builtin_bit_cast(v8bf16, cur.c[m][1]);
        const v8bf16 b1 = __builtin_bit_cast(v8bf16, s_b[rt][m][0][lane]), b2 = __builtin_bit_cast(v8bf16, s_b[rt][m][1][lane]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, m == 0 ? init : acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
      }
      const unsigned o1 = __float_as_uint(M1[rt]), o2 = __float_as_uint(M2[rt]);
#pragma unroll
      for (int r = 0; r < 16; r++) rank1(M1[rt], M2[rt], M3[rt], acc[r], r);
      whose(rt, o1, o2, t);
    }
    cur = nxt;
  }
  }
  RPROF(1)
  __syncthreads();  // nobody reads the operand images any more: their bytes take the waves' partial results
  // -> centroid indices; lanes l and l ^ 32 (same row, the other half of every tile's centroids) merged; one triple per wave and row
#pragma unroll
  for (int rt = 0; rt < NRT; rt++) {
    const int q1 = (int)(__float_as_uint(M1[rt]) & 15u), q2 = (int)(__float_as_uint(M2[rt]) & 15u);
    int j1 = J1[rt] * 32 + 8 * (q1 >> 2) + 4 * h + (q1 & 3), j2 = J2[rt] * 32 + 8 * (q2 >> 2) + 4 * h + (q2 & 3);
    float a1 = M1[rt], a2 = M2[rt], a3 = M3[rt];
    rank_merge3(a1, a2, a3, j1, j2, __shfl_xor(M1[rt], 32, 64), __shfl_xor(M2[rt], 32, 64), __shfl_xor(M3[rt], 32, 64),
                __shfl_xor(j1, 32, 64), __shfl_xor(j2, 32, 64));
    if (lane < 32) {
      s_m[wv][rt * 32 + row][0] = a1;
      s_m[wv][rt * 32 + row][1] = a2;
      s_m[wv][rt * 32 + row][2] = a3;
      s_j[wv][rt * 32 + row][0] = j1;
      s_j[wv][rt * 32 + row][1] = j2;
    }
  }
  __syncthreads();
  RPROF(2)
  // ---- one thread per row (waves 0 and 1): certify, or name the row's two candidates, or leave it to the exact sweep ----
  // the bound's scale: the largest of the stratum's RANK_SLOTS words (as integers: a NaN pattern wins and nothing is certified)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned o = __shfl_xor(bu, off, 64);
    bu = o > bu ? o : bu;
  }
  const float bnd = __uint_as_float(bu);
  const float thr_x = c1 * 1.000001f * __builtin_sqrtf(bnd) * 1.000001f, thr_0 = c2 * bnd + 1e-30f;
  int label = -1, kind = 0;  // kind: 0 certified, 1 settled between two candidates, 2 swept exactly
  const long long rme = r0 + tid;
  const bool mine = tid < RW && rme < st.n_b;
  float xq = 0.f;
  if (mine) {
    float a1 = s_m[0][tid][0], a2 = s_m[0][tid][1], a3 = s_m[0][tid][2];
    int j1 = s_j[0][tid][0], j2 = s_j[0][tid][1];
#pragma unroll
    for (int w = 1; w < NW; w++) rank_merge3(a1, a2, a3, j1, j2, s_m[w][tid][0], s_m[w][tid][1], s_m[w][tid][2], s_j[w][tid][0], s_j[w][tid][1]);
    xq = (s_x[0][tid] + s_x[1][tid]) * 1.000001f;  // (two partial fma chains instead of one: the factor covers the difference)
    const float thr = thr_x * __builtin_sqrtf(xq) + thr_0;
    // (scores can only be infinite -- and their index bits turn an infinity into a signalling NaN -- if the operands are this large)
    const bool tame = xq * bnd < 1e36f;
    if (tame && 2.f * (a1 - a2) > thr && j1 < st.K) {  // d = -2 s; every NaN: the comparison is false
      label = j1;
    } else if (tame && 2.f * (a1 - a3) > thr && j1 != j2 && j1 < st.K && j2 < st.K) {
      kind = 1;
      const int e = atomicAdd(&s_ntwo, 1);
      s_two[e][0] = tid;
      s_two[e][1] = j1 < j2 ? j1 : j2;
      s_two[e][2] = j1 < j2 ? j2 : j1;
    } else {
      kind = 2;
      s_full[atomicAdd(&s_nfull, 1)] = tid;
    }
  }
  // the rows seen so far bound the centroids to come (see the header): the rows' waves publish their largest ||x||^2
  if (wv < RW / 64) {
    unsigned mu = __float_as_uint(mine ? xq * 1.00001f : 0.f);  // (non-negative or NaN: ordered like its bit pattern)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned o = __shfl_xor(mu, off, 64);
      mu = o > mu ? o : mu;
    }
    if (lane == 0 && mu > bu) atomicMax((unsigned *)&bndw[(blockIdx.x * (RW / 64) + wv) & (RANK_SLOTS - 1)], mu);
  }
  __syncthreads();
  RPROF(3)
  // ---- rows between two candidates: the contract's two f64 chains, one wave per row (lane = feature for the loads; the chains
  // themselves are sequential: every lane walks them over the wave's LDS strip) ----
  const int ntwo = s_ntwo;
  for (int e = wv; e < ntwo; e += NW) {
    const int lr = s_two[e][0], lo = s_two[e][1], hi = s_two[e][2];
    const long long r = r0 + lr, f = st.rows ? (long long)st.rows[r] : r;
    const int kc = lane < D ? lane : D - 1;
    const double xv = (double)X[f * ldx + kc], lv = st.C[(size_t)lo * D + kc], hv = st.C[(size_t)hi * D + kc];
    const double cslo = st.csq[lo], cshi = st.csq[hi];
    s_t[wv][0][lane] = xv;
    s_t[wv][1][lane] = lv;
    s_t[wv][2][lane] = hv;
    __builtin_amdgcn_wave_barrier();  // (the strip is the wave's own: LDS accesses of one wave complete in order)
    double dl = 0.0, dh = 0.0;
    for (int k = 0; k + 1 < D; k += 2) {
      const double2 tx = *(const double2 *)&s_t[wv][0][k], tl = *(const double2 *)&s_t[wv][1][k], th = *(const double2 *)&s_t[wv][2][k];
      dl = __builtin_fma(tl.x, tx.x, dl);
      dh = __builtin_fma(th.x, tx.x, dh);
      dl = __builtin_fma(tl.y, tx.y, dl);
      dh = __builtin_fma(th.y, tx.y, dh);
    }
    if (D & 1) {
      dl = __builtin_fma(s_t[wv][1][D - 1], s_t[wv][0][D - 1], dl);
      dh = __builtin_fma(s_t[wv][2][D - 1], s_t[wv][0][D - 1], dh);
    }
    __builtin_amdgcn_wave_barrier();
    dl = __builtin_fma(-2.0, dl, cslo);
    dh = __builtin_fma(-2.0, dh, cshi);
    if (lane == 0) {
      const int lab = dl <= dh ? lo : (dh < dl ? hi : -1);  // first strict minimum over ascending j; a NaN: to the exact sweep
      s_lab[lr] = lab;
      if (lab < 0) s_full[atomicAdd(&s_nfull, 1)] = lr;
    }
  }
  __syncthreads();
  // ---- rows the ranking could not narrow down: the contract's distances to EVERY centroid on the f64 matrix unit, 16 rows at a
  // time -- k_step_assign's arithmetic and orders (centroid fragments as the A operand, rows as B, d = fma(-2, chain, csq), first
  // minimum: register-local per lane, then over the four lane groups, then over the waves) ----
  const int nfull = s_nfull;
  if (nfull > 0) {
    constexpr int KSM = 4 * KB;  // k-steps this instantiation can need (KS = ceil(D / 4) <= KSM)
    const int njt16 = (st.K + 15) >> 4;
    for (int g0 = 0; g0 < nfull; g0 += 16) {
      const int li = g0 + (lane & 15);
      const int lr = s_full[li < nfull ? li : g0];
      const long long r = r0 + lr, f = st.rows ? (long long)st.rows[r] : r;
      double a[KSM];
#pragma unroll
      for (int ks = 0; ks < KSM; ks++) {
        const int k = ks * 4 + (lane >> 4);
        const double v = (double)X[f * ldx + (k < D ? k : D - 1)];  // (unconditional: features past D meet zero fragments or no MFMA at all)
        a[ks] = k < D ? v : 0.0;
      }
      double bd = inf;
      int bi = 0x7fffffff;
      for (int jt = wv; jt < njt16; jt += NW) {
        v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
        double cs[4];
#pragma unroll
        for (int q = 0; q < 4; q++) cs[q] = st.csq[jt * 16 + (lane >> 4) + 4 * q];
        double bq[KSM];
#pragma unroll
        for (int ks = 0; ks < KSM; ks++) bq[ks] = st.Bpack[((size_t)jt * KS + (ks < KS ? ks : KS - 1)) * 64 + lane];
#pragma unroll
        for (int ks = 0; ks < KSM; ks++)
          if (ks < KS) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bq[ks], a[ks], acc, 0, 0, 0);  // (wave-uniform)
        const int j0 = jt * 16 + (lane >> 4);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const double d = __builtin_fma(-2.0, acc[q], cs[q]);
          const bool lt = d < bd;  // a lane's centroids ascend (q, then tile), so strict < keeps its lowest j
          bd = lt ? d : bd;
          bi = lt ? j0 + 4 * q : bi;
        }
      }
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {  // lanes l, l^16, l^32, l^48 hold the same row
        const double d2 = __shfl_xor(bd, off, 64);
        const int j2 = __shfl_xor(bi, off, 64);
        const bool take = (d2 < bd) || (d2 == bd && j2 < bi);
        bd = take ? d2 : bd;
        bi = take ? j2 : bi;
      }
      if (lane < 16) {
        s_fd[wv][lane] = bd;
        s_fi[wv][lane] = bi;
      }
      __syncthreads();
      if (tid < 16 && g0 + tid < nfull) {
        double d = s_fd[0][tid];
        int j = s_fi[0][tid];
#pragma unroll
        for (int w = 1; w < NW; w++) {
          const bool take = (s_fd[w][tid] < d) || (s_fd[w][tid] == d && s_fi[w][tid] < j);
          d = take ? s_fd[w][tid] : d;
          j = take ? s_fi[w][tid] : j;
        }
        s_lab[s_full[g0 + tid]] = j == 0x7fffffff ? 0 : j;  // all-NaN row: label 0 like the oracle
      }
      __syncthreads();
    }
  }
  if (mine && kind != 0) {
    kind = s_lab[tid] >= 0 && kind == 1 ? 1 : 2;  // (a two-candidate row whose chains were not comparable went to the sweep)
    label = s_lab[tid];
  }
  RPROF(4)
  if (mine) {
    st.labels[rme] = label;
    // (result unused: a non-returning atomic -- the workgroup does not wait for it)
    __hip_atomic_fetch_or(&st.mbits[(size_t)label * sdesc.wstride + (rme >> 6)], 1ull << (rme & 63), __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT);
  }
  if (wv < RW / 64 && rk.stat) {  // (the rows' threads sit in the first RW / 64 waves: one add per kind and wave)
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int n = __popcll(__ballot(mine && kind == q));
      if (lane == 0 && n) atomicAdd(rk.stat + q, (unsigned long long)n);
    }
  }
  if constexpr (NRT == 4) {
    // the batch rows as f64 for the M-step (rows of 4 KS doubles, the batch padded to whole 64-row tiles), from the values the
    // prologue loaded.  LAST: stores count in vmcnt, and issued in the prologue every later wait of the wave for a load also waited
    // for their acknowledgements (the sweep of a copying wave took 3 us longer).  Here nothing waits for them but the end of the kernel.
    const int ldg = 4 * KS;
    const long long n_pad = (st.n_b + 63) & ~63LL;
    const int mpar = wv / NRT;
    if (st.Xg && rmine < n_pad) {  // (null: the M-step fetches member rows from the frames, HAMSM_STEP_DIRECT)
      GAS double *xg = st.Xg + (size_t)rmine * ldg;
#pragma unroll
      for (int mm = 0; mm < KBH; mm++)
#pragma unroll
        for (int w = 0; w < 4; w++)
          if (const int k = 16 * (2 * mm + mpar) + 8 * h + 2 * w; k < ldg)  // (ldg is a multiple of 4, k even: the pair stays inside the row)
            *(GAS v2f64 *)(xg + k) = (v2f64){(double)raw[0][mm][2 * w], (double)raw[0][mm][2 * w + 1]};
    }
  }
  asm volatile("" ::"v"(pfv[0]), "v"(pfv[1]));  // the touched lines are only wanted in L2
  RPROF(5)
  RPROF_END
  if (stamp && threadIdx.x == 0) atomicMax(&stamp[1], (unsigned long long)wall_clock64());
}

// M-step: one wave per centroid.  Round trip 1 fetches the centroid's membership bitmap (one word per lane), its count
// and its row; the set bits ARE the members in ascending sample order (no list to sort, no cap on the cluster size):
// each lane expands its word into the wave's LDS strip at the position given by a prefix sum of the popcounts.
// Round trip 2 fetches up to 64 member rows at once (lane = feature); they are added in sample order
// (_k_means_minibatch.pyx:59-108).
#define UPD_LIST 1024
#define UPD_SC 768  // member rows the strip holds per pass: ints [0, 768); weights of a 64-member chunk: ints [768, 896)
// (`apk` != null -- the ranked E-step's run, below: the wave also rewrites its centroid's bf16 fragments, KB blocks of 16 slots)
#define UPD_LROWS 4096  // batch rows whose row list a workgroup of the M-step keeps in LDS (LROWS)
template <typename XT, int H, bool BYVAL, int UPD_NBK, bool DIRECT = false, bool LROWS = false>
__device__ __forceinline__ void step_update_body(const StepDev *__restrict__ steps, const StepPack &pk, int64_t ldx, int D, int KS,
                                                 GAS unsigned short *apk, GAS float *hcs, int KB) {
  SPROF_ID
  SPROF(1, 0)
  const StepDev sdesc = load_step<BYVAL>(steps, pk, blockIdx.y);
  const StepG st(sdesc);
  if (blockIdx.x * 4 >= st.K) return;
  extern __shared__ int lds_i[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int *list = lds_i + wv * UPD_LIST;
  unsigned *srow = (unsigned *)list;
  double *swl = (double *)(list + UPD_SC);
  const GAS double *Xg = st.Xg;
  const int ldg = 4 * KS;  // rows of the batch copy are padded to whole k-steps
  // LROWS: the call's row list (batch row -> frame) comes into LDS while the bitmap words are on their way -- the translation of
  // the members' batch rows was a dependent L2 round trip of its own between the bitmap and the member rows (one per M-step: ~1 us of
  // a 999-step chain).  Requested first, so that waiting for it does not wait for anything younger; one barrier, before any wave leaves.
  int *lrows = lds_i + 4 * UPD_LIST;
  if constexpr (LROWS) {
    int rv[UPD_LROWS / 256];
    const int nb = (int)st.n_b;
#pragma unroll
    for (int q = 0; q < UPD_LROWS / 256; q++) {
      const int e = q * 256 + (int)threadIdx.x;
      rv[q] = (st.rows && e < nb) ? st.rows[e] : e;
    }
#pragma unroll
    for (int q = 0; q < UPD_LROWS / 256; q++) lrows[q * 256 + threadIdx.x] = rv[q];
  }
  const int j = blockIdx.x * 4 + wv;
  if constexpr (!LROWS) {
    if (j >= st.K) return;
  }
  const bool live = j < st.K;
  const int jc = live ? j : st.K - 1;
  const int nw = (int)((st.n_b + 63) >> 6);
  GAS unsigned long long *mb = st.mbits + (size_t)jc * sdesc.wstride;
  const int hv = *gp(sdesc.halt);
  const unsigned long long w0 = mb[lane];  // (wstride is a multiple of 64: always inside the centroid's row)
  const double cnt0 = st.counts[jc];
  double c0[H];
#pragma unroll
  for (int h = 0; h < H; h++) {
    const int k = lane + 64 * h;
    c0[h] = st.C[(size_t)jc * D + (k < D ? k : D - 1)];
  }
  SPROF_W(1, 1)
  if constexpr (LROWS) {
    __syncthreads();
    if (!live) return;
  }
  if (st.idx >= hv) return;  // the stratum's chain stopped before this step
  const bool weighted = st.wf != nullptr;
  // c = c*count; c += x*w in sample order; count += wsum; c *= 1/count  (pyx:80-104)
  double acc[H];
#pragma unroll
  for (int h = 0; h < H; h++) acc[h] = c0[h] * cnt0;
  double wsum = 0.0;
  long long total = 0;
  // The centroid's bitmap row is read 64 words (4096 batch rows) per lane-load; up to UPD_NBK such blocks are taken TOGETHER: their
  // loads are in flight at once, their members go to the strip in one order (block, lane, bit = sample order) and the member rows
  // of all of them are then fetched together -- a batch of 12 500 rows (C5) used to walk its four blocks one dependent round trip
  // after the other, bitmap words and member rows each.  (UPD_NBK = 1 for batches of up to 4096 rows: C3's step is 0.3 us longer
  // with the grouped form's bookkeeping.)
  for (int wb0 = 0; wb0 < nw; wb0 += 64 * UPD_NBK) {
    const int nblk = (nw - wb0 + 63) / 64 < UPD_NBK ? (nw - wb0 + 63) / 64 : UPD_NBK;  // (wave-uniform; >= 1)
    unsigned long long w[UPD_NBK];
#pragma unroll
    for (int bk = 0; bk < UPD_NBK; bk++) {
      w[bk] = 0ull;
      if (bk < nblk) w[bk] = (wb0 == 0 && bk == 0) ? w0 : mb[wb0 + 64 * bk + lane];
    }
    int excl[UPD_NBK], base[UPD_NBK], blk = 0;
#pragma unroll
    for (int bk = 0; bk < UPD_NBK; bk++) {
      excl[bk] = base[bk] = 0;
      if (bk < nblk) {
        if (w[bk]) mb[wb0 + 64 * bk + lane] = 0ull;  // ready for the stratum's next step
        const int pc = __popcll(w[bk]);
        // exclusive prefix sum of the popcounts over the lanes, level by level: sum_{l < lane} pc_l = sum_k #{l < lane: pc_l > k}.
        // A cluster's members are a handful of set bits in the centroid's bitmap words: one or two ballots instead of six
        // LDS-crossbar shuffles in a dependent chain (round 5: this phase was 1.8 of the M-step wave's 5.9 us at C5's shape)
        int ex = 0, tot = 0;
        for (int k = 0;; k++) {
          const unsigned long long bal = __ballot(pc > k);
          if (!bal) break;
          ex += __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          tot += __popcll(bal);
        }
        excl[bk] = ex;
        base[bk] = blk;
        blk += tot;
      }
    }
    SPROF_W(1, 2)
    for (int p0 = 0; p0 < blk; p0 += UPD_SC) {
#pragma unroll
      for (int bk = 0; bk < UPD_NBK; bk++) {  // this lane's members with index in [p0, p0 + SC) go to the strip, in order
        unsigned long long t = w[bk];
        int idx = base[bk] + excl[bk] - p0;
        while (t) {
          const int bit = __builtin_ctzll(t);
          t &= t - 1;
          if (idx >= 0 && idx < UPD_SC) srow[idx] = (unsigned)((wb0 + 64 * bk + lane) * 64 + bit);
          idx++;
        }
      }
      const int np = blk - p0 < UPD_SC ? blk - p0 : UPD_SC;
      if constexpr (DIRECT) {
        // batch rows -> frame indices, the whole strip in one round trip (the call's row list is L2-warm: the E-step read it)
        if constexpr (LROWS) {
          for (int e = lane; e < np; e += 64) srow[e] = (unsigned)lrows[srow[e]];
        } else if (st.rows)
          for (int e = lane; e < np; e += 64) srow[e] = (unsigned)st.rows[srow[e]];
      }
      // member rows fetched together per pass: 64 / H (f64 copy) -- or 32 from the frames: with 32 rows in flight the wave fits
      // 128 registers, four waves per SIMD, and all ~630 workgroups of a C5 launch are resident at once (with 252 they were not:
      // the launch's last workgroups started 7 us after its first)
      constexpr int MW = DIRECT ? 32 / H : 64 / H;
      for (int m0 = 0; m0 < np; m0 += MW) {
        const int cm = np - m0 < MW ? np - m0 : MW;
        if (weighted) {
          if (lane < cm) {
            const long long r = srow[m0 + lane];
            swl[lane] = st.wf[(!DIRECT && st.rows) ? (long long)st.rows[r] : r];
          }
          for (int q = 0; q < cm; q++) wsum = wsum + swl[q];  // sample order (pyx:80-85)
        }
        if constexpr (DIRECT) {
          const GAS XT *Xf = (const GAS XT *)st.X;
          if (cm <= 8 / H) upd_add_direct<8 / H, H, XT>(acc, Xf, ldx, srow + m0, swl, weighted, 0, cm, D, lane);
          else if (cm <= 16 / H) upd_add_direct<16 / H, H, XT>(acc, Xf, ldx, srow + m0, swl, weighted, 0, cm, D, lane);
          else upd_add_direct<32 / H, H, XT>(acc, Xf, ldx, srow + m0, swl, weighted, 0, cm, D, lane);
        } else {
          if (cm <= 16 / H) upd_add_listed<16 / H, H>(acc, Xg, ldg, srow + m0, swl, weighted, 0, cm, D, lane);
          else if (cm <= 32 / H) upd_add_listed<32 / H, H>(acc, Xg, ldg, srow + m0, swl, weighted, 0, cm, D, lane);
          else upd_add_listed<64 / H, H>(acc, Xg, ldg, srow + m0, swl, weighted, 0, cm, D, lane);
        }
      }
    }
    total += blk;
  }
  SPROF_W(1, 3)
  if (total == 0) return;  // no member: wsum = 0, centre unchanged (pyx:105-108)
  if (!weighted) wsum = (double)total;  // the sum of `total` ones is exactly that
  if (!(wsum > 0)) return;  // pyx:87 / :105-108: centre unchanged
  SPROF_W(1, 4)
  const double cnt1 = cnt0 + wsum;
  const double alpha = 1.0 / cnt1;
  double v[H];
#pragma unroll
  for (int h = 0; h < H; h++) {
    int k = lane + 64 * h;
    v[h] = acc[h] * alpha;
    if (k < D) {
      st.C[(size_t)j * D + k] = v[h];
      st.Bpack[((size_t)(j >> 4) * KS + (k >> 2)) * 64 + (k & 3) * 16 + (j & 15)] = v[h];
    }
  }
  if (lane == 0) st.counts[j] = cnt1;
  // ||c||^2 as an fma chain over ascending k (oracle sqnorm_chain): the new row goes through the wave's LDS strip and
  // comes back as broadcast reads, two features at a time (a v_readlane loop costs ~100 cycles per feature here)
  double *vb = (double *)list;
#pragma unroll
  for (int h = 0; h < H; h++) vb[lane + 64 * h] = v[h];  // (lanes past D hold 0.0)
  double a = 0.0;
  for (int k = 0; k + 1 < D; k += 2) {
    const double2 t = *(const double2 *)&vb[k];
    a = __builtin_fma(t.x, t.x, a);
    a = __builtin_fma(t.y, t.y, a);
  }
  if (D & 1) a = __builtin_fma(vb[D - 1], vb[D - 1], a);
  if (lane == 0) st.csq[j] = a;
  if constexpr (H == 1) {
    if (apk) rank_store_row(apk, hcs, KB, j, lane, D, v[0], a);
  }
  SPROF_W(1, 5)
#if defined(STEP_PROFILE) && STEP_PROFILE < 2
  SPROF(1, 5)
#endif
  SPROF_FLUSH(1)
}
template <typename XT, int H, bool BYVAL, int UPD_NBK, bool DIRECT = false, bool LROWS = false>
__global__ __launch_bounds__(256) void k_step_update(const StepDev *__restrict__ steps, const StepPack pk, int64_t ldx, int D,
                                                     int KS) {
  step_update_body<XT, H, BYVAL, UPD_NBK, DIRECT, LROWS>(steps, pk, ldx, D, KS, nullptr, nullptr, 0);
}
template <typename XT, int UPD_NBK, bool DIRECT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DIRECT ? 3 : 1))) void k_step_update_ranked(const StepPack pk, const RankPack rk, int64_t ldx, int D, int KS, int KB) {
  step_update_body<XT, 1, true, UPD_NBK, DIRECT>(nullptr, pk, ldx, D, KS, (GAS unsigned short *)gp(rk.apk[blockIdx.y]), gp(rk.hcs[blockIdx.y]), KB);
}


// block-wide reductions over 256 threads (4 waves); every thread gets the result
__device__ __forceinline__ double block_max_f64(double v, double *sh /*[4]*/) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    double o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  v = sh[0];
#pragma unroll
  for (int w = 1; w < 4; w++) v = sh[w] > v ? sh[w] : v;
  return v;
}
__device__ __forceinline__ int block_sum_i32(int v, int *sh /*[4]*/) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// the low-count test of a stratum's LAST step when the host predicted that no count would be low (every other step's
// test rides on the next E-step): count < ratio * max(count)  (_kmeans.py:1645)
__global__ __launch_bounds__(256) void k_step_check(const StepDev *__restrict__ steps, double ratio, int n_steps) {
  const StepDev sdesc = load_step<false>(steps, StepPack{}, blockIdx.x);
  const StepG st(sdesc);
  if (!(st.flags & 1) || sdesc.perm || st.idx >= *gp(sdesc.halt)) return;
  __shared__ double shd[4];
  __shared__ int shi[4];
  double mx = -__builtin_huge_val();
  for (int j = threadIdx.x; j < st.K; j += 256) mx = st.counts[j] > mx ? st.counts[j] : mx;
  mx = block_max_f64(mx, shd);
  const double thr = ratio * mx;
  int low = 0;
  for (int j = threadIdx.x; j < st.K; j += 256) low += st.counts[j] < thr;
  low = block_sum_i32(low, shi);
  if (threadIdx.x == 0 && low) {
    st.report[1] = low;
    *gp(sdesc.halt) = n_steps;  // nothing left to skip; the reason tells the host to finish the step
    *gp(sdesc.reason) = HAMSM_STOP_LOW;
  }
}

// Re-seeding of low-count centroids on the device (_kmeans.py:1643-1673), for steps that carry the head of the host's
// `permutation(n_b)`: to_reassign = counts < ratio * max; the i-th of them (ascending centroid index) becomes batch
// row perm[i]; their counts become min(counts[~to_reassign]).  One workgroup per step.  When the prediction behind
// the permutation fails (nothing low, or more than the rule / the supplied head allows) the stratum's chain stops.
template <typename XT, bool BYVAL>
__device__ __forceinline__ void step_reassign_body(const StepDev *__restrict__ steps, const StepPack &pk, int64_t ldx, int D, int KS,
                                                   double ratio, GAS unsigned short *apk, GAS float *hcs, int KB) {
  const StepDev sdesc = load_step<BYVAL>(steps, pk, blockIdx.x);
  const StepG st(sdesc);
  if (!(st.flags & 1) || !sdesc.perm || st.idx >= *gp(sdesc.halt)) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double inf = __builtin_huge_val();
  __shared__ double shd[4];
  __shared__ int shi[4];
  __shared__ int lj[256], lr[256];
  __shared__ __attribute__((aligned(16))) double rowbuf[4][512];  // rows of more than 128 features, per wave
  double mx = -inf;
  for (int j = tid; j < st.K; j += 256) mx = st.counts[j] > mx ? st.counts[j] : mx;
  mx = block_max_f64(mx, shd);
  const double thr = ratio * mx;
  int low = 0;
  double mn = -inf;  // max of the negated = min over the centroids that stay
  for (int j = tid; j < st.K; j += 256) {
    double cj = st.counts[j];
    if (cj < thr) low++;
    else mn = -cj > mn ? -cj : mn;
  }
  low = block_sum_i32(low, shi);
  const double new_count = -block_max_f64(mn, shd);  // :1672 np.min(weight_sums[~to_reassign])
  if (tid == 0) st.report[1] = low;
  int why = 0;
  if (low == 0) why = HAMSM_STOP_UNUSED;
  else if (low > sdesc.n_perm || (double)low > 0.5 * (double)st.n_b) why = HAMSM_STOP_CAP;  // :1647
  if (why) {
    if (tid == 0) {
      *gp(sdesc.halt) = st.idx + 1;
      *gp(sdesc.reason) = why;
    }
    return;
  }
  const GAS int32_t *perm = gp(sdesc.perm);
  const GAS XT *X = (const GAS XT *)st.X;
  int base = 0;
  for (int j0 = 0; j0 < st.K; j0 += 256) {
    const int j = j0 + tid;
    const bool flag = j < st.K && st.counts[j] < thr;
    const unsigned long long bal = __ballot(flag);
    const int in_wave = __popcll(bal & ((1ull << lane) - 1));
    __syncthreads();
    if (lane == 0) shi[wv] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      before += w < wv ? shi[w] : 0;
      total += shi[w];
    }
    if (flag) {
      const int e = before + in_wave;
      lj[e] = j;
      lr[e] = perm[base + e];  // :1655 choice(n_b, replace=False, size=n) == permutation(n_b)[:n]
    }
    __syncthreads();
    for (int e = wv; e < total; e += 4) {  // one wave per re-seeded centroid
      const int jc = lj[e];
      const long long r = lr[e];
      const long long f = st.rows ? (long long)st.rows[r] : r;
      if (D > 128) {  // (generic form: rows longer than two features per lane; 512 at a time through the wave's LDS strip)
        double a = 0.0;
        for (int c0 = 0; c0 < D; c0 += 512) {
          const int nc = D - c0 < 512 ? D - c0 : 512;
          double vk[8];
#pragma unroll
          for (int i = 0; i < 8; i++) {
            const int k = c0 + lane + 64 * i;
            vk[i] = (double)X[f * ldx + (k < D ? k : D - 1)];
          }
#pragma unroll
          for (int i = 0; i < 8; i++) {
            const int k = c0 + lane + 64 * i;
            if (k < D) {
              st.C[(size_t)jc * D + k] = vk[i];  // :1667
              st.Bpack[((size_t)(jc >> 4) * KS + (k >> 2)) * 64 + (k & 3) * 16 + (jc & 15)] = vk[i];
              rowbuf[wv][lane + 64 * i] = vk[i];
            }
          }
          __builtin_amdgcn_wave_barrier();  // (the strip is the wave's own: LDS accesses of one wave complete in order)
          if (lane == 0) {
            int i = 0;
            for (; i + 1 < nc; i += 2) {
              const double2 t = *(const double2 *)&rowbuf[wv][i];
              a = __builtin_fma(t.x, t.x, a);
              a = __builtin_fma(t.y, t.y, a);
            }
            if (i < nc) a = __builtin_fma(rowbuf[wv][i], rowbuf[wv][i], a);
          }
          __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) {
          st.csq[jc] = a;
          st.counts[jc] = new_count;
        }
        continue;
      }
      double v[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int k = lane + 64 * h;
        v[h] = (k < D) ? (double)X[f * ldx + k] : 0.0;
        if (k < D) {
          st.C[(size_t)jc * D + k] = v[h];  // :1667
          st.Bpack[((size_t)(jc >> 4) * KS + (k >> 2)) * 64 + (k & 3) * 16 + (jc & 15)] = v[h];
        }
      }
      double a = 0.0;
      for (int k = 0; k < D; k++) {
        double vk = readlane_f64(v[k < 64 ? 0 : 1], k & 63);
        a = __builtin_fma(vk, vk, a);
      }
      if (lane == 0) {
        st.csq[jc] = a;
        st.counts[jc] = new_count;
      }
      if (apk) rank_store_row(apk, hcs, KB, jc, lane, D, v[0], a);  // (the ranked E-step's run: D <= 64, one feature per lane)
    }
    base += total;
    __syncthreads();
  }
}
template <typename XT, bool BYVAL>
__global__ __launch_bounds__(256) void k_step_reassign(const StepDev *__restrict__ steps, const StepPack pk, int64_t ldx, int D,
                                                       int KS, double ratio) {
  step_reassign_body<XT, BYVAL>(steps, pk, ldx, D, KS, ratio, nullptr, nullptr, 0);
}
template <typename XT>
__global__ __launch_bounds__(256) void k_step_reassign_ranked(const StepPack pk, const RankPack rk, int64_t ldx, int D, int KS, double ratio,
                                                              int KB) {
  step_reassign_body<XT, true>(nullptr, pk, ldx, D, KS, ratio, (GAS unsigned short *)gp(rk.apk[blockIdx.x]), gp(rk.hcs[blockIdx.x]), KB);
}

// ---- more than 128 features: generic training kernels ---------------------------------------------------
// k_step_assign / k_step_update keep a row's features in registers as KS <= 32 k-steps and hand the batch to the M-step
// as an f64 copy; past D = 128 neither fits.  These forms take KS = ceil(D / 4) at run time (hamsm_set_geometry):
// same descriptors, same report / halt / reason protocol, same arithmetic and orders (MFMA k-ascending dot products,
// d = fma(-2, dot, csq), first minimum; c*count, += x*w in sample order, *1/count; ||c||^2 as an ascending fma chain),
// so the results are bit-identical to the oracle's as well.  Correct first, not speed-of-light: DESIGN.md §5.6.
//
// E-step: one workgroup per 16 batch rows, the stratum's centroid tiles dealt over its 4 waves, operands straight from L2.
template <typename XT, int RT, bool STAGE>
__global__ __launch_bounds__(256) void k_step_assign_generic(const StepDev *__restrict__ steps, int64_t ldx, int D, int KS,
                                                             double ratio) {
  static_assert(STAGE || RT == 1, "several row tiles share the staged rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char xs_raw[];  // STAGE: the workgroup's rows, [16 RT][DS] XT
  const StepDev sdesc = load_step<false>(steps, StepPack{}, blockIdx.y);
  const StepG st(sdesc);
  constexpr int RW = 16 * RT;  // rows per workgroup
  const long long r0 = (long long)blockIdx.x * RW;
  if (r0 >= st.n_b) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kq = lane >> 4;
  const double inf = __builtin_huge_val();
  __shared__ double shd[4];
  __shared__ int shi[4];
  if (blockIdx.x == 0 && (st.flags & 6) && st.idx < *gp(sdesc.halt)) {
    // the tests that ride on the E-step (see k_step_assign): low counts of the stratum's previous step (flags & 4,
    // _kmeans.py:1645), a zero count before this one (flags & 2, :2037); either finding stops the chain here
    int z = 0;
    double mx = -inf;
    for (int j = tid; j < st.K; j += 256) {
      const double cj = st.counts[j];
      z |= (cj == 0.0);
      mx = cj > mx ? cj : mx;
    }
    int any_low = 0;
    if (st.flags & 4) {
      mx = block_max_f64(mx, shd);
      const double thr = ratio * mx;
      int low = 0;
      for (int j = tid; j < st.K; j += 256) low += st.counts[j] < thr;
      low = block_sum_i32(low, shi);
      if (tid == 0 && low) gp(sdesc.prev_report)[1] = low;
      any_low = low;
    }
    const int any_z = __syncthreads_or((st.flags & 2) ? z : 0);
    if (tid == 0) {
      if (any_low) {
        *gp(sdesc.halt) = st.idx;
        *gp(sdesc.reason) = HAMSM_STOP_LOW;
      } else if (any_z) {
        st.report[0] = 1;
        *gp(sdesc.halt) = st.idx;
        *gp(sdesc.reason) = HAMSM_STOP_ZERO;
      }
    }
  }
  // centroids are the MFMA's A operand, 16 batch rows its B operand: lane l accumulates, for batch row l&15 of a row tile,
  // the centroids (l>>4) + 4q of a centroid tile.  Wave w owns the centroid tiles w, w + 4, ...: two at a time against the
  // workgroup's RT row tiles (every fragment of Bpack it loads serves RT MFMAs), 8 k-steps of operands in flight
  long long f[RT];
  bool ok[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    const long long r = r0 + rt * 16 + (lane & 15);
    ok[rt] = r < st.n_b;
    const long long rc = ok[rt] ? r : st.n_b - 1;  // (rows past the batch repeat its last row; nothing of them is kept)
    f[rt] = st.rows ? (long long)st.rows[rc] : rc;
  }
  const GAS XT *xr = (const GAS XT *)st.X + f[0] * ldx;  // (!STAGE: RT == 1)
  const int DS = D | 1;  // odd row stride: the 16 rows of a fragment read fall into different banks
  XT *xs = (XT *)xs_raw;
  if constexpr (STAGE) {
    // all four waves need all rows (each sweeps its own centroid tiles): gathered once, coalesced (wave w: rows 4 RT w ...,
    // a lane per feature), 8 loads in flight per lane
    constexpr int RPW = RW / 4;
#pragma unroll 1
    for (int rr = 0; rr < RPW; rr++) {
      const int row = wv * RPW + rr, tile = row >> 4;
      long long fsel = f[0];
#pragma unroll
      for (int t = 1; t < RT; t++) fsel = tile == t ? f[t] : fsel;
      const long long fr = __shfl(fsel, row & 15, 64);  // (lane l holds the frame of row l & 15 of every tile)
      const GAS XT *src = (const GAS XT *)st.X + fr * ldx;
      for (int k0 = 0; k0 < D; k0 += 64 * 8) {
        XT v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int k = k0 + lane + 64 * i;
          v[i] = src[k < D ? k : D - 1];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int k = k0 + lane + 64 * i;
          if (k < D) xs[row * DS + k] = v[i];
        }
      }
    }
    __syncthreads();
  }
  const XT *xl = xs + (lane & 15) * DS;
  const int njt = (st.K + 15) >> 4;
  double bd[RT];
  int bj[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    bd[rt] = inf;
    bj[rt] = 0x7fffffff;
  }
  for (int jt0 = wv; jt0 < njt; jt0 += 8) {
    const bool two = jt0 + 4 < njt;
    const GAS double *bp0 = st.Bpack + (size_t)jt0 * KS * 64 + lane;
    const GAS double *bp1 = st.Bpack + (size_t)(two ? jt0 + 4 : jt0) * KS * 64 + lane;
    v4f64 acc0[RT], acc1[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) acc0[rt] = acc1[rt] = (v4f64){0.0, 0.0, 0.0, 0.0};
    constexpr int U = 8;  // k-steps of operands in flight (measured at D = 816, K_s = 100, RT = 1: 8 -> 39 us per launch, 16 -> 46 us)
    for (int ks0 = 0; ks0 < KS; ks0 += U) {
      double xa[RT][U], b0[U], b1[U];
#pragma unroll
      for (int i = 0; i < U; i++) {  // unconditional loads at clamped positions; what lies past KS / D is deselected below
        const int ks = ks0 + i < KS ? ks0 + i : KS - 1, k = ks * 4 + kq;
        const int kc = k < D ? k : D - 1;
        if constexpr (STAGE) {
#pragma unroll
          for (int rt = 0; rt < RT; rt++) xa[rt][i] = (double)xl[rt * 16 * DS + kc];
        } else {
          xa[0][i] = (double)xr[kc];
        }
        b0[i] = bp0[(size_t)ks * 64];
        b1[i] = bp1[(size_t)ks * 64];
      }
#pragma unroll
      for (int i = 0; i < U; i++) {
        const bool in = ks0 + i < KS, inx = in && (ks0 + i) * 4 + kq < D;
        const double v0 = in ? b0[i] : 0.0, v1 = in ? b1[i] : 0.0;
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
          const double a = inx ? xa[rt][i] : 0.0;  // fma(0, 0, acc) == acc
          acc0[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, a, acc0[rt], 0, 0, 0);
          acc1[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, a, acc1[rt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (u == 1 && !two) continue;
      const int j0 = (jt0 + 4 * u) * 16 + kq;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double cq = st.csq[j0 + 4 * q];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
          const double d = __builtin_fma(-2.0, u ? acc1[rt][q] : acc0[rt][q], cq);
          const bool lt = d < bd[rt];  // a lane's centroids ascend (q, then tile), so strict < keeps its lowest j
          bd[rt] = lt ? d : bd[rt];
          bj[rt] = lt ? j0 + 4 * q : bj[rt];
        }
      }
    }
  }
  __shared__ double sd[4][RW];
  __shared__ int sj[4][RW];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    double d = bd[rt];
    int j = bj[rt];
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {  // lanes l, l^16, l^32, l^48 hold the same batch row
      const double d2 = __shfl_xor(d, off, 64);
      const int j2 = __shfl_xor(j, off, 64);
      const bool take = (d2 < d) || (d2 == d && j2 < j);
      d = take ? d2 : d;
      j = take ? j2 : j;
    }
    if (lane < 16) {
      sd[wv][rt * 16 + lane] = d;
      sj[wv][rt * 16 + lane] = j;
    }
  }
  __syncthreads();
  if (tid < RW && r0 + tid < st.n_b) {
    double d = sd[0][tid];
    int j = sj[0][tid];
#pragma unroll
    for (int w = 1; w < 4; w++) {
      const double d2 = sd[w][tid];
      const int j2 = sj[w][tid];
      const bool take = (d2 < d) || (d2 == d && j2 < j);
      d = take ? d2 : d;
      j = take ? j2 : j;
    }
    if (j == 0x7fffffff) j = 0;  // all-NaN row: label 0 like the oracle
    const long long rr = r0 + tid;
    st.labels[rr] = j;
    __hip_atomic_fetch_or(&st.mbits[(size_t)j * sdesc.wstride + (rr >> 6)], 1ull << (rr & 63), __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT);
  }
}

// M-step: one workgroup per centroid, a thread per feature (1024 features per pass over the member list, 4 per thread).
// Wave 0 expands the centroid's membership bitmap, 64 words at a time, into an LDS list of frame indices (and weights);
// all threads then add the members in sample order, reading the frames where they lie (no f64 copy of the batch).
#define UPDG_BLK 4096
template <typename XT>
__global__ __launch_bounds__(256) void k_step_update_generic(const StepDev *__restrict__ steps, int64_t ldx, int D, int KS) {
  const StepDev sdesc = load_step<false>(steps, StepPack{}, blockIdx.y);
  const StepG st(sdesc);
  const int j = blockIdx.x;
  if (j >= st.K || st.idx >= *gp(sdesc.halt)) return;  // (block-uniform)
  __shared__ int sfr[UPDG_BLK];     // frame index of the block's m-th member
  __shared__ __attribute__((aligned(16))) double swl[UPDG_BLK];  // its weight
  __shared__ int s_blk;
  __shared__ double s_wsum;
  const int tid = threadIdx.x, lane = tid & 63;
  const int nw = (int)((st.n_b + 63) >> 6);
  GAS unsigned long long *mb = st.mbits + (size_t)j * sdesc.wstride;
  const bool weighted = st.wf != nullptr;
  const GAS XT *X = (const GAS XT *)st.X;
  const double cnt0 = st.counts[j];
  double wsum = 0.0, cnt1 = 0.0, alpha = 0.0;
  long long total = 0;
  const int nfc = (D + 1023) / 1024;
  for (int fc = 0; fc < nfc; fc++) {
    const bool first = fc == 0, last = fc == nfc - 1;
    // c = c*count; c += x*w in sample order; count += wsum; c *= 1/count  (_k_means_minibatch.pyx:80-104)
    double acc[4];
    int kk[4];
#pragma unroll
    for (int h = 0; h < 4; h++) {
      const int k = fc * 1024 + tid + 256 * h;
      kk[h] = k < D ? k : D - 1;  // (threads past D carry a copy that is never stored)
      acc[h] = st.C[(size_t)j * D + kk[h]] * cnt0;
    }
    for (int wb = 0; wb < nw; wb += 64) {  // (wstride is a multiple of 64: the words stay inside the centroid's row)
      __syncthreads();                     // the previous block's list has been consumed
      if (tid < 64) {
        const unsigned long long w = mb[wb + lane];
        if (last && w) mb[wb + lane] = 0ull;  // ready for the stratum's next step
        const int pc = __popcll(w);
        int inc = pc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int t = __shfl_up(inc, off, 64);
          inc += lane >= off ? t : 0;
        }
        unsigned long long t = w;
        int idx = inc - pc;
        while (t) {
          const int bit = __builtin_ctzll(t);
          t &= t - 1;
          sfr[idx++] = (wb + lane) * 64 + bit;  // the batch row, for now
        }
        if (lane == 63) s_blk = inc;
      }
      __syncthreads();
      const int blk = s_blk;
      for (int m = tid; m < blk; m += 256) {
        const long long r = sfr[m];
        const long long f = st.rows ? (long long)st.rows[r] : r;
        sfr[m] = (int)f;
        if (weighted) swl[m] = st.wf[f];
      }
      __syncthreads();
      if (first && weighted && tid == 0)
        for (int m = 0; m < blk; m++) wsum = wsum + swl[m];  // sample order (pyx:80-85)
      for (int m0 = 0; m0 < blk; m0 += 8) {  // 8 member rows in flight
        XT xv[8][4];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const size_t fo = (size_t)sfr[m0 + q < blk ? m0 + q : blk - 1] * ldx;
#pragma unroll
          for (int h = 0; h < 4; h++) xv[q][h] = X[fo + kk[h]];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const double wq = weighted ? swl[m0 + q < blk ? m0 + q : blk - 1] : 1.0;
#pragma unroll
          for (int h = 0; h < 4; h++) {
            const double nv = acc[h] + (double)xv[q][h] * wq;
            acc[h] = (m0 + q < blk) ? nv : acc[h];
          }
        }
      }
      if (first) total += blk;
    }
    if (first) {
      if (total == 0) return;  // no member: centre unchanged (pyx:105-108); nothing to clear either
      if (weighted) {
        if (tid == 0) s_wsum = wsum;
        __syncthreads();
        wsum = s_wsum;
      } else {
        wsum = (double)total;  // the sum of `total` ones is exactly that
      }
      if (!(wsum > 0)) {  // pyx:87 / :105-108: centre unchanged; the bitmap is still to be cleared
        for (int wi = tid; wi < nw; wi += 256) mb[wi] = 0ull;
        return;
      }
      cnt1 = cnt0 + wsum;
      alpha = 1.0 / cnt1;
    }
    if (nfc == 1) __syncthreads();  // (every thread is done with the weights: swl takes the new row for the chain below)
#pragma unroll
    for (int h = 0; h < 4; h++) {
      const int k = fc * 1024 + tid + 256 * h;
      if (k < D) {
        const double v = acc[h] * alpha;
        st.C[(size_t)j * D + k] = v;
        st.Bpack[((size_t)(j >> 4) * KS + (k >> 2)) * 64 + (k & 3) * 16 + (j & 15)] = v;
        if (nfc == 1) swl[k] = v;
      }
    }
  }
  // ||c||^2 as an fma chain over ascending k (oracle sqnorm_chain): the new row goes through LDS (left there by a single
  // pass; else read back 4096 features at a time, one coalesced round trip -- a single thread loading it element by element
  // would expose D memory latencies)
  double a = 0.0;
  for (int c0 = 0; c0 < D; c0 += UPDG_BLK) {
    __syncthreads();  // the workgroup's stores to the row are visible to its loads (same CU); swl is free again
    const int nc = D - c0 < UPDG_BLK ? D - c0 : UPDG_BLK;
    if (nfc > 1) {  // (a single pass left the row in swl already)
      for (int i = tid; i < nc; i += 256) swl[i] = st.C[(size_t)j * D + c0 + i];
      __syncthreads();
    }
    if (tid == 0) {
      int i = 0;
      for (; i + 15 < nc; i += 16) {  // 8 LDS reads in flight ahead of the 16 dependent fmas
        double2 t[8];
#pragma unroll
        for (int q = 0; q < 8; q++) t[q] = *(const double2 *)&swl[i + 2 * q];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          a = __builtin_fma(t[q].x, t[q].x, a);
          a = __builtin_fma(t[q].y, t[q].y, a);
        }
      }
      for (; i < nc; i++) a = __builtin_fma(swl[i], swl[i], a);
    }
  }
  if (tid == 0) {
    st.counts[j] = cnt1;
    st.csq[j] = a;
  }
}

#ifdef HAMSM_EXPERIMENTS
#include "experiments/train_ahead.inc"
#include "experiments/train_ranked.inc"
#include "experiments/train_persist.inc"
#endif


template <typename XT, int KS, int RT, bool BYVAL>
static void launch_step_kernels(hamsm_ctx *c, hipStream_t q, const StepDev *d_steps, const StepPack &pk, int ny,
                                long long max_nb, int maxK, int64_t ldx, double ratio, bool any_perm) {
  constexpr int JB = KS <= 4 ? 8 : (KS <= 8 ? 4 : (KS <= 16 ? 2 : 1));
  dim3 ga((unsigned)((max_nb + 16 * RT - 1) / (16 * RT)), (unsigned)ny);
  // every 128th launch group of the context's own queue is sampled: events around the E-step and the M-step launch, and the
  // E-step's own first-wave-in / last-wave-out stamps (slots prepared in bulk by hamsm_step_kernel_ms(reset))
  const bool sample = q == c->stream && !HAMSM_CAPTURING(c) && c->ks_stamps && (c->ks_tick++ & 127) == 0 && c->ks_pending.size() < HAMSM_KS_SLOTS;
  hamsm_ctx::KSample ks{nullptr, nullptr, nullptr};
  unsigned long long *stamp = nullptr;
  if (sample) {
    stamp = c->ks_stamps + 2 * c->ks_pending.size();
    auto ev = [&]() {
      if (!c->ks_free.empty()) {
        hipEvent_t e = c->ks_free.back();
        c->ks_free.pop_back();
        return e;
      }
      hipEvent_t e = nullptr;
      hipEventCreate(&e);
      return e;
    };
    ks.e0 = ev();
    ks.e1 = ev();
    ks.e2 = ev();
    hipEventRecord(ks.e0, q);
  }
  hipLaunchKernelGGL((k_step_assign<XT, KS, JB, RT, BYVAL>), ga, dim3(256), 0, q, d_steps, pk, ldx, c->D, ratio, stamp);
  if (sample) hipEventRecord(ks.e1, q);
  size_t lds = sizeof(int) * (4 * UPD_LIST);
  dim3 gu((unsigned)((maxK + 3) / 4), (unsigned)ny);
  if (c->cur_direct) {  // member rows from the frames (the E-step wrote no f64 copy)
    if (c->D <= 64) {
      if (max_nb > 4096) hipLaunchKernelGGL((k_step_update<XT, 1, BYVAL, 4, true>), gu, dim3(256), lds, q, d_steps, pk, ldx, c->D, c->KS);
      else if (c->step_lrows && max_nb <= UPD_LROWS)
        hipLaunchKernelGGL((k_step_update<XT, 1, BYVAL, 1, true, true>), gu, dim3(256), lds + sizeof(int) * UPD_LROWS, q, d_steps, pk, ldx, c->D, c->KS);
      else hipLaunchKernelGGL((k_step_update<XT, 1, BYVAL, 1, true>), gu, dim3(256), lds, q, d_steps, pk, ldx, c->D, c->KS);
    } else {
      if (max_nb > 4096) hipLaunchKernelGGL((k_step_update<XT, 2, BYVAL, 4, true>), gu, dim3(256), lds, q, d_steps, pk, ldx, c->D, c->KS);
      else hipLaunchKernelGGL((k_step_update<XT, 2, BYVAL, 1, true>), gu, dim3(256), lds, q, d_steps, pk, ldx, c->D, c->KS);
    }
  } else if (c->D <= 64) {
    if (max_nb > 4096) hipLaunchKernelGGL((k_step_update<XT, 1, BYVAL, 4>), gu, dim3(256), lds, q, d_steps, pk, ldx, c->D, c->KS);
    else hipLaunchKernelGGL((k_step_update<XT, 1, BYVAL, 1>), gu, dim3(256), lds, q, d_steps, pk, ldx, c->D, c->KS);
  } else {
    if (max_nb > 4096) hipLaunchKernelGGL((k_step_update<XT, 2, BYVAL, 4>), gu, dim3(256), lds, q, d_steps, pk, ldx, c->D, c->KS);
    else hipLaunchKernelGGL((k_step_update<XT, 2, BYVAL, 1>), gu, dim3(256), lds, q, d_steps, pk, ldx, c->D, c->KS);
  }
  if (sample) {
    hipEventRecord(ks.e2, q);
    c->ks_pending.push_back(ks);
  }
  if (any_perm)
    hipLaunchKernelGGL((k_step_reassign<XT, BYVAL>), dim3((unsigned)ny), dim3(256), 0, q, d_steps, pk, ldx, c->D, c->KS, ratio);
}


// row tiles per E-step workgroup: as many as the registers allow (RT * KS <= 32) while the launch keeps >= 128 workgroups
template <typename XT, int KS, bool BYVAL>
static void launch_step_rt(hamsm_ctx *c, hipStream_t q, const StepDev *d_steps, const StepPack &pk, int ny, long long max_nb,
                           int maxK, int64_t ldx, double ratio, bool any_perm) {
  const long long tiles = (max_nb + 15) / 16 * ny;
  // measured: C3 (10^4 rows, 4 x 250 centroids, 64 KB of operands per stratum) RT = 1 / 2 / 4 -> 25.0 / 26.5 / 29.0 us
  // per call -- fatter workgroups lose more latency hiding than they save in operand traffic; C5-shaped (5 x 10^4
  // rows, 4 x 1250 centroids x 64 features, 647 KB per stratum) RT = 1 / 2 -> 468 / 420 us -- there the operand
  // stream through L2 binds.  Hence two row tiles once a stratum's operands exceed 256 KB.
  const char *rt_s = getenv("HAMSM_STEP_RT");  // (tests / experiments force a tile count)
  const int rt_env = rt_s ? atoi(rt_s) : 0;
  const size_t operand_bytes = (size_t)((maxK + 15) / 16) * KS * 512;
  const int rt_cap = rt_env ? rt_env : (operand_bytes >= (256u << 10) ? 2 : 1);
  if constexpr (KS <= 8) {
    if (tiles >= 4 * 128 && rt_cap >= 4) return launch_step_kernels<XT, KS, 4, BYVAL>(c, q, d_steps, pk, ny, max_nb, maxK, ldx, ratio, any_perm);
  }
  if constexpr (KS <= 16) {
    if (tiles >= 2 * 128 && rt_cap >= 2) return launch_step_kernels<XT, KS, 2, BYVAL>(c, q, d_steps, pk, ny, max_nb, maxK, ldx, ratio, any_perm);
  }
  launch_step_kernels<XT, KS, 1, BYVAL>(c, q, d_steps, pk, ny, max_nb, maxK, ldx, ratio, any_perm);
}

template <typename XT>
static int launch_steps(hamsm_ctx *c, hipStream_t q, const StepDev *d_steps, const StepPack &pk, int ny, long long max_nb,
                        int maxK, int64_t ldx, double ratio, bool any_perm) {
  if (c->KS > 32) {  // more than 128 features: descriptors always in memory, one queue (run_steps)
    if (!d_steps) return hamsm_fail(c, HAMSM_EINVAL, "generic training kernels take their descriptors from memory");
    // the E-step's rows in LDS while they fit; 2 or 4 row tiles per workgroup (every Bpack fragment then serves 2 / 4 MFMAs)
    // once the launch is large enough
    const size_t row_bytes = sizeof(XT) * 16 * (size_t)(c->D | 1), cap = (size_t)c->generic_stage_kb * 1024;
    const long long tiles = (max_nb + 15) / 16 * ny;
    const char *rt_s = getenv("HAMSM_GENERIC_RT");  // (tests / experiments force a tile count)
    // (measured at D = 816, 4 x 100 centroids, 500 row tiles per launch: RT = 1 / 2 / 4 -> 18.7 / 24.9 / 24.9 ms per pass -- the
    // kernel runs at the f64 MFMA rate there and fewer, fatter workgroups only lose occupancy; hence >= 1024 workgroups kept)
    int rt = rt_s ? atoi(rt_s) : (tiles >= 4 * 1024 ? 4 : (tiles >= 2 * 1024 ? 2 : 1));
    while (rt > 1 && rt * row_bytes > cap) rt >>= 1;
    rt = rt >= 4 ? 4 : (rt >= 2 ? 2 : 1);
    const dim3 ge((unsigned)((max_nb + 16 * rt - 1) / (16 * rt)), (unsigned)ny);
    const void *kfn = (const void *)k_step_assign_generic<XT, 1, false>;
    size_t xs_bytes = 0;
    if (row_bytes <= cap) {
      xs_bytes = rt * row_bytes;
      kfn = rt == 4 ? (const void *)k_step_assign_generic<XT, 4, true>
                    : (rt == 2 ? (const void *)k_step_assign_generic<XT, 2, true> : (const void *)k_step_assign_generic<XT, 1, true>);
      HIPCHK(c, hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xs_bytes));
    }
    {
      int D_ = c->D, KS_ = c->KS;
      void *args[] = {(void *)&d_steps, (void *)&ldx, (void *)&D_, (void *)&KS_, (void *)&ratio};
      HIPCHK(c, hipLaunchKernel(kfn, ge, dim3(256), args, xs_bytes, q));
    }
    hipLaunchKernelGGL(k_step_update_generic<XT>, dim3((unsigned)maxK, (unsigned)ny), dim3(256), 0, q, d_steps, ldx, c->D, c->KS);
    if (any_perm)
      hipLaunchKernelGGL((k_step_reassign<XT, false>), dim3((unsigned)ny), dim3(256), 0, q, d_steps, pk, ldx, c->D, c->KS, ratio);
    return HAMSM_OK;
  }
  switch (c->KS) {
#define CASE(KSV)                                                                               \
  case KSV:                                                                                     \
    if (d_steps) launch_step_rt<XT, KSV, false>(c, q, d_steps, pk, ny, max_nb, maxK, ldx, ratio, any_perm); \
    else launch_step_rt<XT, KSV, true>(c, q, d_steps, pk, ny, max_nb, maxK, ldx, ratio, any_perm);          \
    break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(6) CASE(8) CASE(12) CASE(16) CASE(24) CASE(32)
#undef CASE
    default: return hamsm_fail(c, HAMSM_ENOTSUP, "KS=%d", c->KS);
  }
  return HAMSM_OK;
}


// the ranked form of a launch group (descriptors by value, <= PACK_NY strata): E-step on the bf16 matrix unit, the M-step and the
// device re-seeding in the forms that also maintain the fragments
template <typename XT>
static int launch_steps_ranked(hamsm_ctx *c, hipStream_t q, const StepPack &pk, const RankPack &rk, int ny, long long max_nb, int maxK,
                               int64_t ldx, double ratio, bool any_perm) {
  const int KB = (c->D + 15) / 16;  // blocks of 16 feature slots (the csq term rides in the accumulator's initial value)
  // the light form's certificate (assign.hip, launch_sorted): 3 MFMAs x 17 roundings per 16 slots, taken four times (204 KB);
  // the three dropped 2^-14-level piece products (6186 u, u = 2^-24); rounding of c and of f64 frames to f32 (2 u); the index
  // bits (64 u, 32 u); -csq / 2 rounded to f32 once (1 u)
  const float c1 = 2.02f * (float)(6186 + 2 * (2 + 204 * KB) + 64) / 16777216.0f;
  const float c2 = 2.02f * (float)(204 * KB + 1 + 32) / 16777216.0f;
  // 256-row workgroups where 128-row ones of ALL the context's strata (the queues run side by side) would not fit the chip at one
  // workgroup per CU -- C5's shape: 4 x 98 -> 4 x 49; runs without the f64 batch copy only (the copy is written per 128 rows)
  const bool direct = pk.s[0].Xg == nullptr;
  // (256-row workgroups: built, bit-identical, measured slower at C5's shape -- 54 against 50 us per launch: every phase of a
  // workgroup scales with its rows, the fragment stream does not bind -- and taken only when forced: HAMSM_STEP_WIDE=2)
  const bool wide = direct && c->step_wide == 2;
  const dim3 ga((unsigned)((max_nb + (wide ? 255 : 127)) / (wide ? 256 : 128)), (unsigned)ny);
  const bool sample = q == c->stream && !HAMSM_CAPTURING(c) && c->ks_stamps && (c->ks_tick++ & 127) == 0 && c->ks_pending.size() < HAMSM_KS_SLOTS;
  hamsm_ctx::KSample ks{nullptr, nullptr, nullptr};
  unsigned long long *stamp = nullptr;
  if (sample) {
    stamp = c->ks_stamps + 2 * c->ks_pending.size();
    auto ev = [&]() {
      if (!c->ks_free.empty()) {
        hipEvent_t e = c->ks_free.back();
        c->ks_free.pop_back();
        return e;
      }
      hipEvent_t e = nullptr;
      hipEventCreate(&e);
      return e;
    };
    ks.e0 = ev();
    ks.e1 = ev();
    ks.e2 = ev();
    hipEventRecord(ks.e0, q);
  }
  switch (KB) {
#define CASE(KBV) \
  case KBV:                                                                                                                                 \
    if (wide && c->step_pipe) hipLaunchKernelGGL((k_step_assign_ranked<XT, KBV, 8, RANK_NW, true>), ga, dim3(RANK_NW * 64), 0, q, pk, rk, ldx, c->D, c->KS, ratio, c1, c2, stamp); \
    else if (wide) hipLaunchKernelGGL((k_step_assign_ranked<XT, KBV, 8>), ga, dim3(RANK_NW * 64), 0, q, pk, rk, ldx, c->D, c->KS, ratio, c1, c2, stamp); \
    else if (direct && c->step_wide == 3 && c->step_pipe) hipLaunchKernelGGL((k_step_assign_ranked<XT, KBV, 4, 4, true>), ga, dim3(4 * 64), 0, q, pk, rk, ldx, c->D, c->KS, ratio, c1, c2, stamp); \
    else if (direct && c->step_wide == 3) hipLaunchKernelGGL((k_step_assign_ranked<XT, KBV, 4, 4>), ga, dim3(4 * 64), 0, q, pk, rk, ldx, c->D, c->KS, ratio, c1, c2, stamp); \
    else if (c->step_pipe) hipLaunchKernelGGL((k_step_assign_ranked<XT, KBV, 4, RANK_NW, true>), ga, dim3(RANK_NW * 64), 0, q, pk, rk, ldx, c->D, c->KS, ratio, c1, c2, stamp); \
    else hipLaunchKernelGGL((k_step_assign_ranked<XT, KBV, 4>), ga, dim3(RANK_NW * 64), 0, q, pk, rk, ldx, c->D, c->KS, ratio, c1, c2, stamp);      \
    break;
    CASE(1) CASE(2) CASE(3) CASE(4)
#undef CASE
    default: return hamsm_fail(c, HAMSM_ENOTSUP, "ranked E-step: %d blocks of feature slots", KB);
  }
  if (sample) hipEventRecord(ks.e1, q);
  const size_t lds = sizeof(int) * (4 * UPD_LIST);
  const dim3 gu((unsigned)((maxK + 3) / 4), (unsigned)ny);
  if (direct) {
    if (max_nb > 4096) hipLaunchKernelGGL((k_step_update_ranked<XT, 4, true>), gu, dim3(256), lds, q, pk, rk, ldx, c->D, c->KS, KB);
    else hipLaunchKernelGGL((k_step_update_ranked<XT, 1, true>), gu, dim3(256), lds, q, pk, rk, ldx, c->D, c->KS, KB);
  } else {
    if (max_nb > 4096) hipLaunchKernelGGL((k_step_update_ranked<XT, 4, false>), gu, dim3(256), lds, q, pk, rk, ldx, c->D, c->KS, KB);
    else hipLaunchKernelGGL((k_step_update_ranked<XT, 1, false>), gu, dim3(256), lds, q, pk, rk, ldx, c->D, c->KS, KB);
  }
  if (sample) {
    hipEventRecord(ks.e2, q);
    c->ks_pending.push_back(ks);
  }
  if (any_perm) hipLaunchKernelGGL((k_step_reassign_ranked<XT>), dim3((unsigned)ny), dim3(256), 0, q, pk, rk, ldx, c->D, c->KS, ratio, KB);
  return HAMSM_OK;
}

// steps [0,n): consecutive steps of distinct strata share launches (grid.y)
#ifdef HAMSM_EXPERIMENTS
static int run_steps_experiments(hamsm_ctx *c, const void *X, int dt, int64_t ldx, const int32_t *d_rows, const double *d_w,
                                 const hamsm_step *h_steps, int64_t n_steps, double ratio, const int32_t *h_perm, int64_t n_perm_total,
                                 int32_t *h_report, int64_t *h_resume, int32_t *h_reason, int want_zero_flag,
                                 int32_t **labels_of_first);
#endif
static int run_steps(hamsm_ctx *c, const void *X, int dt, int64_t ldx, const int32_t *d_rows, const double *d_w,
                     const hamsm_step *h_steps, int64_t n_steps, double ratio, const int32_t *h_perm, int64_t n_perm_total,
                     int32_t *h_report, int64_t *h_resume, int32_t *h_reason, int want_zero_flag,
                     int32_t **labels_of_first) {
#ifdef HAMSM_EXPERIMENTS
  // (libhamsm_hip_exp.so only: a context with one of the measured-slower experiments switched on takes the launcher that has
  // them threaded through it -- csrc/experiments/)
  if (c->train_persist || c->ahead || c->train_ranked || c->train_graph)
    return run_steps_experiments(c, X, dt, ldx, d_rows, d_w, h_steps, n_steps, ratio, h_perm, n_perm_total, h_report, h_resume, h_reason,
                                 want_zero_flag, labels_of_first);
#endif
  const size_t esz = dt == HAMSM_F32 ? 4 : 8;
  const bool generic = c->KS > 32;  // more than 128 features (k_step_*_generic)
  timespec ts_in;
  clock_gettime(CLOCK_MONOTONIC, &ts_in);
  int64_t lab_hi = 0, nb_max = 0;
  for (int64_t i = 0; i < n_steps; i++) {
    const hamsm_step &s = h_steps[i];
    if (s.n_b > nb_max) nb_max = s.n_b;
    if (s.stratum < 0 || s.stratum >= c->S) return hamsm_fail(c, HAMSM_EINVAL, "bad stratum in step %lld", (long long)i);
    if (c->st[s.stratum].K == 0) return hamsm_fail(c, HAMSM_EUNFIT, "stratum %d has no centroids", s.stratum);
    if (s.n_b < 0 || s.rows_off < 0) return hamsm_fail(c, HAMSM_EINVAL, "bad step %lld", (long long)i);
    if (s.perm_off >= 0 && (!h_perm || s.n_perm < 0 || s.perm_off + s.n_perm > n_perm_total))
      return hamsm_fail(c, HAMSM_EINVAL, "step %lld: permutation outside h_perm", (long long)i);
    int64_t hi = (d_rows ? s.rows_off : 0) + s.n_b;
    if (hi > lab_hi) lab_hi = hi;
  }
  if (n_steps >= 0x7f000000) return hamsm_fail(c, HAMSM_EINVAL, "too many steps");
  // launch groups = maximal runs of distinct strata
  std::vector<int64_t> gstart;
  {
    std::vector<char> used(c->S);
    int64_t i0 = 0;
    while (i0 < n_steps) {
      std::fill(used.begin(), used.end(), 0);
      int64_t j0 = i0;
      while (j0 < n_steps && !used[h_steps[j0].stratum] && j0 - i0 < 65535) used[h_steps[j0++].stratum] = 1;
      gstart.push_back(i0);
      i0 = j0;
    }
    gstart.push_back(n_steps);
  }
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return o;
  };
  size_t o_steps = take(sizeof(StepDev) * n_steps), o_rep = take(sizeof(int32_t) * 2 * n_steps),
         o_lab = take(sizeof(int32_t) * (size_t)lab_hi), o_halt = take(sizeof(int32_t) * 2 * (size_t)c->S),
         o_perm = take(sizeof(int32_t) * (size_t)(n_perm_total > 0 ? n_perm_total : 1));
  // the gathered batch copy and the membership bitmaps are reused by every launch group: one slot per stratum position,
  // stream-ordered
  const int nslot = c->S < 65535 ? c->S : 65535;
  const size_t xg_row = sizeof(double) * 4 * (size_t)c->KS;
  const size_t nb_pad = ((size_t)nb_max + 63) & ~(size_t)63;  // whole E-step tiles (up to 4 x 16 rows)
  int Kmax = 0;
  for (int s = 0; s < c->S; s++) Kmax = c->st[s].K > Kmax ? c->st[s].K : Kmax;
  // Calls whose E-step is bound by the f64 matrix unit rank their centroids on the bf16 unit instead (k_step_assign_ranked: same
  // labels).  Decided per run -- the fragments are maintained from step to step -- by its heaviest CALL (rows x centroids x
  // features x 2 of one stratum's step: 0.15 GFLOP by default -- tools/rankgate.py: at 0.04 GFLOP, C3, the f64 kernel is 1.3x
  // faster, at 0.16 the two are even, at 0.3 the ranked one is 1.5x faster, at 2 GFLOP, C5, 4.7x); at most 64 features, a planned
  // pass with row lists.  HAMSM_STEP_RANKED = 0 never, 1 by size (default), 2 always (tests).
  const int KBr = (c->D + 15) / 16;
  bool ranked = c->step_ranked && c->assign_filter && !generic && c->KS <= 16 && d_rows && h_report && nslot == c->S;
  if (ranked && c->step_ranked == 1) {
    double top = 0;
    for (int64_t q = 0; q < n_steps; q++) {
      const double flop = 2.0 * (double)h_steps[q].n_b * c->st[h_steps[q].stratum].K * c->D;
      top = flop > top ? flop : top;
    }
    ranked = top >= c->step_ranked_min_flop;
  }
  // member rows straight from the frames (no f64 copy): ranked runs; the f64-MFMA kernels of up to 128 features on request
  const bool direct = ranked ? c->step_direct != 0 : (!generic && c->step_direct_plain != 0);
  c->cur_direct = direct;
  size_t o_xg = take((generic || direct) ? 0 : xg_row * (size_t)nslot * nb_pad);  // (the generic M-step reads the frames in place)
  const int wstride = (int)((((nb_max + 63) >> 6) + 63) & ~(int64_t)63);  // bitmap words per centroid
  const size_t mb_bytes = sizeof(unsigned long long) * (size_t)nslot * Kmax * wstride;
  size_t o_mb = take(mb_bytes), o_tail = take(sizeof(StepDev) * (size_t)c->S);
  const size_t apk_words = (size_t)((Kmax + 31) / 32) * KBr * 2 * 256;  // per stratum
  const size_t hcs_words = (size_t)((Kmax + 31) / 32) * 32;
  size_t o_apk = take(ranked ? sizeof(unsigned) * apk_words * (size_t)c->S : 0), o_bnd = take(ranked ? sizeof(unsigned) * RANK_SLOTS * (size_t)c->S : 0),
         o_hcs = take(ranked ? sizeof(float) * hcs_words * (size_t)c->S : 0), o_rstat = take(sizeof(unsigned long long) * 4);
  int rc = hamsm_ensure_scratch(c, off);
  if (rc) return rc;
  char *b = (char *)c->scratch;
  std::vector<StepDev> sd(n_steps);
  for (int64_t i = 0; i < n_steps; i++) {
    const hamsm_step &s = h_steps[i];
    Stratum &t = c->st[s.stratum];
    StepDev &d = sd[i];
    d.X = (const char *)X + (size_t)s.x_off * ldx * esz;
    d.rows = d_rows ? d_rows + s.rows_off : nullptr;
    d.wf = d_w ? d_w + s.x_off : nullptr;
    d.n_b = s.n_b;
    d.labels = (int32_t *)(b + o_lab) + (d_rows ? s.rows_off : 0);
    d.C = t.C;
    d.counts = t.counts;
    d.csq = t.csq;
    d.Bpack = t.Bpack;
    d.K = t.K;
    d.flags = (s.flags & 1) | ((want_zero_flag && !(s.flags & HAMSM_STEP_FIRST)) ? 2 : 0);
    d.report = (int32_t *)(b + o_rep) + 2 * i;
    d.prev_report = nullptr;
    d.perm = ((s.flags & 1) && s.perm_off >= 0) ? (const int32_t *)(b + o_perm) + s.perm_off : nullptr;
    d.n_perm = s.n_perm;
    d.idx = (int)i;
    d.halt = (int32_t *)(b + o_halt) + s.stratum;
    d.reason = (int32_t *)(b + o_halt) + c->S + s.stratum;
  }
  // the low-count test of a step is evaluated by the E-step of the stratum's next step (the counts do not change in
  // between); only a stratum's last step needs a launch of its own
  std::vector<int64_t> last_of(c->S, -1), tail_checks;
  for (int64_t i = 0; i < n_steps; i++) {
    if (h_steps[i].n_b <= 0) continue;  // launches nothing: cannot evaluate anything either
    const int s = h_steps[i].stratum;
    const int64_t p = last_of[s];
    if (p >= 0 && (sd[p].flags & 1) && !sd[p].perm) {
      sd[i].flags |= 4;
      sd[i].prev_report = sd[p].report;
    }
    last_of[s] = i;
  }
  for (int s = 0; s < c->S; s++)
    if (last_of[s] >= 0 && (sd[last_of[s]].flags & 1) && !sd[last_of[s]].perm) tail_checks.push_back(last_of[s]);
  // member-list slot of a step = its stratum (chains of different strata run concurrently on different queues), else its
  // position inside its launch group
  for (size_t g = 0; g + 1 < gstart.size(); g++)
    for (int64_t j0 = gstart[g]; j0 < gstart[g + 1]; j0++) {
      const size_t slot = nslot == c->S ? (size_t)h_steps[j0].stratum : (size_t)(j0 - gstart[g]);
      sd[j0].Xg = direct ? nullptr : (double *)(b + o_xg + slot * nb_pad * xg_row);
      sd[j0].mbits = (unsigned long long *)(b + o_mb) + slot * Kmax * wstride;
      sd[j0].wstride = wstride;
    }
  if (labels_of_first) *labels_of_first = sd[0].labels;
  HIPCHK(c, hipMemcpyAsync(b + o_steps, sd.data(), sizeof(StepDev) * n_steps, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(b + o_rep, 0, sizeof(int32_t) * 2 * n_steps, c->stream));
  HIPCHK(c, hipMemsetAsync(b + o_halt, 0x7f, sizeof(int32_t) * c->S, c->stream));  // stop words: 0x7f7f7f7f = never
  HIPCHK(c, hipMemsetAsync(b + o_halt + sizeof(int32_t) * c->S, 0, sizeof(int32_t) * c->S, c->stream));
  if (n_perm_total > 0)
    HIPCHK(c, hipMemcpyAsync(b + o_perm, h_perm, sizeof(int32_t) * (size_t)n_perm_total, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(b + o_mb, 0, mb_bytes, c->stream));
  const StepDev *d_steps = (const StepDev *)(b + o_steps);
  const bool dbg_enq = getenv("HAMSM_DEBUG_ENQUEUE") != nullptr;
  hamsm_timer_begin(c);
  if (ranked) {
    // the run's fragments and bounds, from the tables as they stand (whoever changed them between runs -- the host's settling of
    // a stopped chain, a new seeding -- needs no hook)
    HIPCHK(c, hipMemsetAsync(b + o_bnd, 0, sizeof(unsigned) * RANK_SLOTS * (size_t)c->S, c->stream));
    HIPCHK(c, hipMemsetAsync(b + o_rstat, 0, sizeof(unsigned long long) * 4, c->stream));
    for (int s = 0; s < c->S; s++) {
      const Stratum &t = c->st[s];
      if (t.K > 0)
        hipLaunchKernelGGL(k_rank_pack, dim3((unsigned)((t.K + 31) / 32)), dim3(64), 0, c->stream, (const double *)t.C, (const double *)t.csq, t.K,
                           c->D, KBr, (unsigned *)(b + o_apk) + apk_words * s, (float *)(b + o_hcs) + hcs_words * s,
                           (unsigned *)(b + o_bnd) + RANK_SLOTS * s);
    }
  }
  // lanes: stratum s always runs on queue s % n_lanes, so every stratum's chain stays ordered; a lane's part of a
  // launch group travels by value (<= PACK_NY strata) -- otherwise one queue and descriptors in memory
  int n_lanes = c->step_lanes;
  for (size_t g = 0; g + 1 < gstart.size() && n_lanes > 1; g++) {
    std::vector<int> cnt(n_lanes, 0);
    for (int64_t q = gstart[g]; q < gstart[g + 1]; q++)
      if (++cnt[h_steps[q].stratum % n_lanes] > PACK_NY) n_lanes = 1;
  }
  if (gstart.size() <= 2 || c->S < 2 || nslot != c->S || !d_rows || generic) n_lanes = 1;  // (labels scratch is per step only with row lists)
  auto lane_of = [&](int64_t q) { return n_lanes > 1 ? h_steps[q].stratum % n_lanes : 0; };
  hipStream_t lanes[4] = {c->stream, nullptr, nullptr, nullptr};
  if (n_lanes > 1) {
    if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    for (int l = 1; l < n_lanes; l++) {
      if (!c->side_stream[l - 1]) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->side_stream[l - 1], hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[l - 1], hipEventDisableTiming));
      }
      lanes[l] = c->side_stream[l - 1];
      HIPCHK(c, hipStreamWaitEvent(lanes[l], c->ev_fork, 0));
    }
  }
  StepPack pk;
  memset(&pk, 0, sizeof(pk));
  timespec ts0, ts1;
  if (dbg_enq) clock_gettime(CLOCK_MONOTONIC, &ts0);
  // Bounded run-ahead (MFMA-bound calls only).  The host enqueues a launch group in ~7 us; where the device needs ~60 us for it
  // the whole pass is in the queues long before its first steps have run, and a chain that stops -- at C5's shape the fused first
  // steps stop every stratum, every pass -- left thousands of launches to drain (16 ms per 1000-call run even as near-empty
  // kernels).  So the enqueue waits, every RA_W groups, until the device is at most 2 RA_W groups behind (events per queue: no
  // bubble, >= RA_W groups are still queued), reads the stop words through a queue of its own, drops stopped strata from the
  // launches that follow and ends the run when no live stratum is left.  Latency-bound calls (C3: 17 us of device work per
  // group, the host barely ahead, stops rare) enqueue as before.
  const bool bounded = ranked && h_report && h_resume && gstart.size() > 2 * 16 + 1;
  constexpr size_t RA_W = 16;
  std::vector<char> halted(c->S, 0);
  std::vector<int64_t> last_group(c->S, -1);
  if (bounded) {
    for (size_t g = 0; g + 1 < gstart.size(); g++)
      for (int64_t q = gstart[g]; q < gstart[g + 1]; q++) last_group[h_steps[q].stratum] = (int64_t)g;
    if (!c->poll_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->poll_stream, hipStreamNonBlocking));
    if (c->h_poll_n < (size_t)c->S) {
      if (c->h_poll) hipHostFree(c->h_poll);
      c->h_poll = nullptr;
      c->h_poll_n = 0;
      HIPCHK(c, hipHostMalloc((void **)&c->h_poll, sizeof(int32_t) * (size_t)c->S));
      c->h_poll_n = (size_t)c->S;
    }
    for (int l = 0; l < n_lanes; l++)
      for (int k = 0; k < 4; k++)
        if (!c->ev_ring[l][k]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_ring[l][k], hipEventDisableTiming));
  }
  // Latency-bound passes on two queues (C3: 2 x 2 000 launches, 3.6 us of host time each, against 16.9 ms of device time per
  // pass): each queue is fed by a thread of its own, so that neither chain ever waits for the other's enqueue (one thread
  // alternating between them stayed 1-4 ms ahead of the device at best).  Descriptors travel by value; the threads share nothing
  // they write (launch sampling belongs to the context's own queue: the calling thread).
  const bool threaded = c->enqueue_threads > 1 && !ranked && !bounded && !generic && n_lanes >= 2 && gstart.size() > 64 && !HAMSM_CAPTURING(c);
  if (threaded) {
    auto enqueue_lane = [&](int l) -> int {
      StepPack pk2;
      memset(&pk2, 0, sizeof(pk2));
      std::vector<int64_t> mine2;
      for (size_t g = 0; g + 1 < gstart.size(); g++) {
        const int64_t i = gstart[g], j = gstart[g + 1];
        long long max_nb = 0;
        int maxK = 0;
        bool any_perm = false;
        mine2.clear();
        for (int64_t q = i; q < j; q++) {
          if (lane_of(q) != l) continue;
          mine2.push_back(q);
          any_perm = any_perm || sd[q].perm;
          if (h_steps[q].n_b > max_nb) max_nb = h_steps[q].n_b;
          if (c->st[h_steps[q].stratum].K > maxK) maxK = c->st[h_steps[q].stratum].K;
        }
        const int ny = (int)mine2.size();
        if (ny == 0 || max_nb <= 0) continue;
        if (ny > PACK_NY) return HAMSM_EINVAL;  // (cannot happen: n_lanes is chosen so that a lane's strata fit)
        for (int q = 0; q < ny; q++) pk2.s[q] = sd[mine2[q]];
        pk2.n_next = 0;
        if (g + 2 < gstart.size()) {  // the same strata of the next group, in the same positions
          int nn = 0;
          for (int64_t q = j; q < gstart[g + 2] && nn < ny; q++) {
            if (lane_of(q) != l) continue;
            pk2.nx[nn++] = StepNext{sd[q].X, sd[q].rows, sd[q].n_b};
          }
          pk2.n_next = nn;
        }
        const int r = (dt == HAMSM_F32) ? launch_steps<float>(c, lanes[l], nullptr, pk2, ny, max_nb, maxK, ldx, ratio, any_perm)
                                        : launch_steps<double>(c, lanes[l], nullptr, pk2, ny, max_nb, maxK, ldx, ratio, any_perm);
        if (r) return r;
      }
      return hipGetLastError() == hipSuccess ? HAMSM_OK : HAMSM_EDEVICE;  // (the launch error state is per thread)
    };
    int rcl[4] = {HAMSM_OK, HAMSM_OK, HAMSM_OK, HAMSM_OK};
    std::thread side[3];
    bool started[3] = {false, false, false};
    for (int l = 1; l < n_lanes; l++) {
      try {
        side[l - 1] = std::thread([&, l]() {
          if (hipSetDevice(c->device) != hipSuccess) {
            rcl[l] = HAMSM_EDEVICE;
            return;
          }
          rcl[l] = enqueue_lane(l);
        });
        started[l - 1] = true;
      } catch (...) {  // (no thread to be had: this one feeds the queue too -- nothing leaves through the C ABI)
      }
    }
    rc = enqueue_lane(0);
    for (int l = 1; l < n_lanes; l++) {
      if (started[l - 1]) side[l - 1].join();
      else rcl[l] = enqueue_lane(l);
      if (!rc) rc = rcl[l];
    }
    if (rc) {
      hipDeviceSynchronize();  // (all lanes: nothing may still read `sd` once it is gone)
      return rc == HAMSM_EDEVICE ? hamsm_fail(c, HAMSM_EDEVICE, "step launches (threaded enqueue)") : rc;
    }
  }
  std::vector<int64_t> mine;
  for (size_t g = 0; !threaded && g + 1 < gstart.size(); g++) {
    const int64_t i = gstart[g], j = gstart[g + 1];
    if (bounded && g > 0 && g % RA_W == 0) {
      const size_t k = g / RA_W;  // windows 0 .. k - 1 are enqueued
      for (int l = 0; l < n_lanes; l++) HIPCHK(c, hipEventRecord(c->ev_ring[l][(k - 1) & 3], lanes[l]));
      if (k >= 2) {
        for (int l = 0; l < n_lanes; l++) HIPCHK(c, hipEventSynchronize(c->ev_ring[l][(k - 2) & 3]));
        HIPCHK(c, hipMemcpyAsync(c->h_poll, b + o_halt, sizeof(int32_t) * (size_t)c->S, hipMemcpyDeviceToHost, c->poll_stream));
        HIPCHK(c, hipStreamSynchronize(c->poll_stream));
        bool any_live = false;
        for (int s = 0; s < c->S; s++) {
          halted[s] = c->h_poll[s] < (int32_t)(n_steps < 0x7f000000 ? n_steps : 0x7f000000);
          any_live = any_live || (!halted[s] && last_group[s] >= (int64_t)g);
        }
        if (!any_live) break;  // every chain with steps left has stopped: the host settles them, the next run resumes
      }
    }
    for (int l = 0; l < n_lanes; l++) {
      long long max_nb = 0;
      int maxK = 0;
      bool any_perm = false;
      mine.clear();
      for (int64_t q = i; q < j; q++) {
        if (lane_of(q) != l) continue;
        if (halted[h_steps[q].stratum]) continue;  // (its steps from here on would only drain)
        mine.push_back(q);
        any_perm = any_perm || sd[q].perm;
        if (h_steps[q].n_b > max_nb) max_nb = h_steps[q].n_b;
        if (c->st[h_steps[q].stratum].K > maxK) maxK = c->st[h_steps[q].stratum].K;
      }
      const int ny = (int)mine.size();
      if (ny == 0 || max_nb <= 0) continue;
      if (ranked) {
        for (size_t c0_ = 0; c0_ < mine.size() && !rc; c0_ += PACK_NY) {  // (descriptors by value: PACK_NY strata per launch)
          const int nyc = (int)(mine.size() - c0_ < PACK_NY ? mine.size() - c0_ : PACK_NY);
          RankPack rk;
          memset(&rk, 0, sizeof(rk));
          long long mnb = 0;
          int mK = 0;
          bool ap = false;
          for (int q = 0; q < nyc; q++) {
            const int64_t sq = mine[c0_ + q];
            const int y = h_steps[sq].stratum;
            pk.s[q] = sd[sq];
            rk.apk[q] = (unsigned *)(b + o_apk) + apk_words * y;
            rk.hcs[q] = (float *)(b + o_hcs) + hcs_words * y;
            rk.bnd[q] = (unsigned *)(b + o_bnd) + RANK_SLOTS * y;
            ap = ap || sd[sq].perm;
            mnb = h_steps[sq].n_b > mnb ? h_steps[sq].n_b : mnb;
            mK = c->st[y].K > mK ? c->st[y].K : mK;
          }
          rk.stat = (unsigned long long *)(b + o_rstat);
          pk.n_next = 0;
          if (g + 2 < gstart.size()) {  // the same strata's steps of the next group, in the same positions (none: n_b = 0)
            for (int q = 0; q < nyc; q++) {
              pk.nx[q] = StepNext{nullptr, nullptr, 0};
              for (int64_t q2 = j; q2 < gstart[g + 2]; q2++)
                if (h_steps[q2].stratum == h_steps[mine[c0_ + q]].stratum) pk.nx[q] = StepNext{sd[q2].X, sd[q2].rows, sd[q2].n_b};
            }
            pk.n_next = nyc;
          }
          if (mnb <= 0) continue;
          rc = (dt == HAMSM_F32) ? launch_steps_ranked<float>(c, lanes[l], pk, rk, nyc, mnb, mK, ldx, ratio, ap)
                                 : launch_steps_ranked<double>(c, lanes[l], pk, rk, nyc, mnb, mK, ldx, ratio, ap);
        }
        if (rc) {
          hipDeviceSynchronize();
          return rc;
        }
        continue;
      }
      // (one lane: `mine` is the whole group, in order; several lanes: at most PACK_NY strata each, by construction of n_lanes)
      const StepDev *dptr = d_steps + i;
      if (ny <= PACK_NY && !generic) {
        dptr = nullptr;  // descriptors by value
        for (int q = 0; q < ny; q++) pk.s[q] = sd[mine[q]];
        pk.n_next = 0;
        if (g + 2 < gstart.size()) {
          // the same strata of the next group, in the same positions
          int nn = 0;
          for (int64_t q = j; q < gstart[g + 2] && nn < ny; q++) {
            if (lane_of(q) != l) continue;
            pk.nx[nn++] = StepNext{sd[q].X, sd[q].rows, sd[q].n_b};
          }
          pk.n_next = nn;
        }
      }
      rc = (dt == HAMSM_F32) ? launch_steps<float>(c, lanes[l], dptr, pk, ny, max_nb, maxK, ldx, ratio, any_perm)
                             : launch_steps<double>(c, lanes[l], dptr, pk, ny, max_nb, maxK, ldx, ratio, any_perm);
      if (rc) {
        hipDeviceSynchronize();  // (all lanes: nothing may still read `sd` once it is gone)
        return rc;
      }
    }
  }
  if (dbg_enq) {
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    fprintf(stderr, "[hamsm] enqueued %zu launch groups on %d queue(s) in %.3f ms (%.3f ms of preparation before the first launch)\n",
            gstart.size() - 1, n_lanes, (ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-6,
            (ts0.tv_sec - ts_in.tv_sec) * 1e3 + (ts0.tv_nsec - ts_in.tv_nsec) * 1e-6);
  }
  for (int l = 1; l < n_lanes; l++) {
    HIPCHK(c, hipEventRecord(c->ev_join[l - 1], lanes[l]));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[l - 1], 0));
  }
  if (!tail_checks.empty()) {
    // (staged in a buffer the context owns: the asynchronous upload must not read from a local)
    c->h_stage.resize(sizeof(StepDev) * tail_checks.size());
    StepDev *tails = (StepDev *)c->h_stage.data();
    for (size_t k = 0; k < tail_checks.size(); k++) tails[k] = sd[tail_checks[k]];
    HIPCHK(c, hipMemcpyAsync(b + o_tail, tails, sizeof(StepDev) * tail_checks.size(), hipMemcpyHostToDevice, c->stream));
    // (one launch, a workgroup per stratum: they used to be one launch each, at the very end of the chain)
    hipLaunchKernelGGL(k_step_check, dim3((unsigned)tail_checks.size()), dim3(256), 0, c->stream, (const StepDev *)(b + o_tail), ratio, (int)n_steps);
  }
  hamsm_timer_end(c, ST_STEP);
  if (hipError_t le = hipGetLastError(); le != hipSuccess) {
    hipDeviceSynchronize();
    return hamsm_fail(c, HAMSM_EDEVICE, "step launches: %s", hipGetErrorString(le));
  }
  if (!h_report) {  // (the stepwise entry point: one step, no stop protocol)
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the descriptor upload reads from the local `sd`
    return HAMSM_OK;
  }
  // where every chain stands: stop words + reasons, then the per-step reports
  std::vector<int32_t> hw(2 * (size_t)c->S);
  {
    hipError_t e = hipMemcpyAsync(hw.data(), b + o_halt, sizeof(int32_t) * 2 * c->S, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_report, b + o_rep, sizeof(int32_t) * 2 * n_steps, hipMemcpyDeviceToHost, c->stream);
    unsigned long long rstat[4] = {0, 0, 0, 0};
    if (ranked && e == hipSuccess) e = hipMemcpyAsync(rstat, b + o_rstat, sizeof(rstat), hipMemcpyDeviceToHost, c->stream);
    hipError_t es = hipStreamSynchronize(c->stream);  // always: `sd`, `hw` and `rstat` are locals the copies above read / write
    if (ranked) {
      for (int k = 0; k < 3; k++) c->rank_stat[k] += rstat[k];
      c->rank_stat[3]++;
    }
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "minibatch_run read-back: %s", hipGetErrorString(e));
  }
  if (dbg_enq) {
    timespec ts_out;
    clock_gettime(CLOCK_MONOTONIC, &ts_out);
    fprintf(stderr, "[hamsm] run_steps: %.3f ms from entry to the results on the host\n",
            (ts_out.tv_sec - ts_in.tv_sec) * 1e3 + (ts_out.tv_nsec - ts_in.tv_nsec) * 1e-6);
  }
  if (h_resume && h_reason) {
    // stop word -> first step of the stratum that was not applied
    for (int s = 0; s < c->S; s++) {
      h_reason[s] = hw[c->S + s];
      h_resume[s] = n_steps;
    }
    for (int64_t i = n_steps - 1; i >= 0; i--) {
      const int s = h_steps[i].stratum;
      if (i >= hw[s]) h_resume[s] = i;
    }
  }
  return HAMSM_OK;
}
#ifdef HAMSM_EXPERIMENTS
#include "experiments/run_steps_experiments.inc"
#endif


// Not part of the product ABI (test / bench hook): out[4] = batch rows the ranked E-step certified / settled between two candidates /
// swept exactly, and the runs that took it, since the last reset; mode >= 0 also sets HAMSM_STEP_RANKED's value for the context
// (test hook) shape of the ranked training kernels: wide = 0 never / 1 by size / 2 always 256-row E-step workgroups; direct = 1: the
// M-step fetches member rows from the frames, 0: round 4's f64 batch copy.  Negative: leave as is.
extern "C" int hamsm_debug_step_shape(hamsm_ctx *c, int wide, int direct) {
  CTX(c);
  if (wide >= 0) c->step_wide = wide > 3 ? 3 : wide;
  if (direct >= 0) c->step_direct = direct != 0;
  return HAMSM_OK;
}
// (test hook) the f64-MFMA training kernels: plain = 1 member rows from the frames / 0 the f64 batch copy; lrows = 1 the call's row
// list in LDS (batches of <= 4096 rows).  Negative: leave as is.
extern "C" int hamsm_debug_step_plain(hamsm_ctx *c, int plain, int lrows) {
  CTX(c);
  if (plain >= 0) c->step_direct_plain = plain != 0;
  if (lrows >= 0) c->step_lrows = lrows != 0;
  return HAMSM_OK;
}
extern "C" int hamsm_debug_rank_stats(hamsm_ctx *c, unsigned long long *out, int reset, int mode) {
  CTX(c);
  for (int k = 0; k < 4; k++) {
    if (out) out[k] = c->rank_stat[k];
    if (reset) c->rank_stat[k] = 0;
  }
  if (mode >= 0) c->step_ranked = mode > 2 ? 2 : mode;
  return HAMSM_OK;
}

extern "C" int hamsm_minibatch_run(hamsm_ctx *c, const void *X, int dt, int64_t ldx, const int32_t *d_rows, const double *d_w,
                                   const hamsm_step *h_steps, int64_t n_steps, double ratio, const int32_t *h_perm,
                                   int64_t n_perm_total, int32_t *h_report, int64_t *h_resume, int32_t *h_reason) {
  CTX(c);
  if (c->st.empty()) return hamsm_fail(c, HAMSM_EINVAL, "set_geometry first");
  if (!h_report || !h_resume || !h_reason) return hamsm_fail(c, HAMSM_EINVAL, "h_report / h_resume / h_reason missing");
  for (int s = 0; s < c->S; s++) {
    h_resume[s] = n_steps;
    h_reason[s] = HAMSM_STOP_NONE;
  }
  if (n_steps <= 0) return HAMSM_OK;
  return run_steps(c, X, dt, ldx, d_rows, d_w, h_steps, n_steps, ratio, h_perm, n_perm_total, h_report, h_resume, h_reason, 1,
                   nullptr);
}

extern "C" int hamsm_minibatch_step(hamsm_ctx *c, int s, const void *X, int dt, int64_t ldx, const int32_t *rows, int64_t n_b,
                                    const double *wf, int32_t *labels_out) {
  CTX(c);
  if (c->st.empty() || s < 0 || s >= c->S) return hamsm_fail(c, HAMSM_EINVAL, "bad stratum");
  if (c->st[s].K == 0) return hamsm_fail(c, HAMSM_EUNFIT, "stratum %d has no centroids", s);
  if (n_b <= 0) return HAMSM_OK;
  // the row list becomes the step's d_rows (offset 0): labels scratch is indexed by batch position
  hamsm_step st{s, 0, 0, 0, n_b, -1, 0, 0};
  int32_t *lab = nullptr;
  int rc = run_steps(c, X, dt, ldx, rows, wf, &st, 1, 0.0, nullptr, 0, nullptr, nullptr, nullptr, 0, &lab);
  if (rc) return rc;
  if (labels_out) HIPCHK(c, hipMemcpyAsync(labels_out, lab, sizeof(int32_t) * n_b, hipMemcpyDeviceToDevice, c->stream));
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
  CTX(c);
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

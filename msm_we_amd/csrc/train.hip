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
#include "common.h"

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
  size_t o_rg = take(sizeof(long long) * 2 * G), o_cr = take(sizeof(int32_t) * G * c->S),
         o_cnt = take(sizeof(long long) * G * c->S), o_st = take(sizeof(long long) * G * c->S),
         o_rank = take(sizeof(int32_t) * (size_t)hi);
  int rc = hamsm_ensure_scratch(c, off);
  if (rc) return rc;
  char *b = (char *)c->scratch;
  HIPCHK(c, hipMemcpyAsync(b + o_rg, h_range, sizeof(long long) * 2 * G, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b + o_cr, h_cr, sizeof(int32_t) * G * c->S, hipMemcpyHostToDevice, c->stream));
  hamsm_timer_begin(c);
  if (nmax <= BR_WAVE_MAX)
    hipLaunchKernelGGL(c->S <= BR_SREG ? k_build_rows_wave<true> : k_build_rows_wave<false>, dim3((unsigned)((G + 3) / 4)), dim3(256),
                       sizeof(int) * 8 * c->S, c->stream, bin, (const long long *)(b + o_rg), (const int32_t *)(b + o_cr), c->S,
                       (long long)G, rows, (long long *)(b + o_cnt), (long long *)(b + o_st));
  else
    hipLaunchKernelGGL(k_build_rows, dim3((unsigned)G), dim3(BR_T), 0, c->stream, bin, (const long long *)(b + o_rg),
                       (const int32_t *)(b + o_cr), c->S, (int32_t *)(b + o_rank), rows, (long long *)(b + o_cnt),
                       (long long *)(b + o_st));
  hamsm_timer_end(c, ST_PREPASS);
  HIPCHK(c, hipGetLastError());
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
  double *G;            // [n][n] Gram matrix of the batch (k_seed_g only)
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
  double *crow = sm;                         // [L][D] candidate rows
  double *part = crow + (size_t)L * D;       // [L][SEED_T] per-thread partial potentials
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
#define SEED_G_LDS(LF) ((size_t)(LF) * SEED_PL + (LF) * 32 + 16 + SEED_NLDS + 2 * 64 * SEED_PST + SEED_NCH)
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
      // exclusive running sum of the chunk totals, formed by every lane (the reads are LDS broadcasts); lane q keeps base q
      double tv[SEED_NCH];
#pragma unroll
      for (int q = 0; q < SEED_NCH; q++) tv[q] = lbase[q];
      double run = 0.0, mybase = 0.0;
#pragma unroll
      for (int q = 0; q < SEED_NCH; q++) {
        mybase = lane == q ? run : mybase;
        run = run + tv[q];
      }
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
    size_t XbT, w, xsq, cl, cum, cand, ct, u, ch, sub;
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
    it.G = fast ? (double *)(b + o_G + g_bytes * (size_t)(i % g_slots)) : nullptr;
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
    int Lmax = 1;
    for (int i = 0; i < n_items; i++) Lmax = h_L[i] > Lmax ? h_L[i] : Lmax;
    size_t lds = sizeof(double) * (size_t)Lmax * ((size_t)D + SEED_T + SEED_T / SEED_GROUP + 1);
    if (lds > 160 * 1024 - 256)
      return hamsm_fail(c, HAMSM_ENOTSUP, "k-means++ seeding of a batch of more than %d rows with %d candidates of %d features needs %zu B of LDS",
                        SEED_NLDS, Lmax, D, lds);
    HIPCHK(c, hipFuncSetAttribute((const void *)k_seed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_seed, dim3((unsigned)n_items), dim3(SEED_T), lds, c->stream, (const SeedItem *)(b + o_items), D);
  }
  HIPCHK(c, hipGetLastError());
  for (int i = 0; i < n_items; i++) {
    int s = h_stratum[i];
    Stratum &t = c->st[s];
    t.K = h_K[i];
    HIPCHK(c, hipMemsetAsync(t.counts, 0, sizeof(double) * t.K, c->stream));  // _kmeans.py:2271
    rc = hamsm_pack_stratum(c, s, nullptr);
    if (rc) return rc;
  }
  c->table_dirty = true;
  hamsm_timer_end(c, ST_SEED);
  // h_u / h_sub were pageable host memory: hipMemcpyAsync staged them before returning, no sync needed.
  c->seed_chosen_dev = b + lay[0].ch;  // hamsm_kmeans_seed reads item 0's choices
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
struct SuspEntry {  // look-ahead blocks: batch row r is nearest to j1, j2 or j3 (-1: none), whichever the centroids of its step say
  int r, j1, j2, j3;
};
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
// look-ahead blocks (k_step_assign_ahead / k_step_update_ahead): what a step of a block carries besides its StepDev
// (kept apart: the step-by-step kernels take their descriptors by value and every byte of them is on the launch path)
struct AheadDev {
  int pos, pad_;              // position of the step among its stratum's steps of the block (0 = the block's first); < 0: not in a block
  double *slot;               // [AHEAD_SLOTS] largest centroid displacement this step's M-step measured, spread over AHEAD_SLOTS words
                              // (wave j adds to word j % AHEAD_SLOTS: 250 atomic maxima on ONE word cost 10 us); readers take their maximum
  const double *prev_slot;    // the slots of the stratum's previous step (null: this step's labels are exact, nothing to check)
  const double *ref_slot;     // the slots this step's bound is derived from (null: bound 0)
  double ref_scale;           //   bound = ref_scale * *ref_slot
  double *bound;              // [1] the bound the E-step used (read back by the M-step that checks it)
  // the step's M-step reads the stratum's tables where its StepDev points (C, counts, csq) and writes ALL of them, moved or
  // not, to these: the E-step of a later iteration may be reading the old ones in the same launch (two buffers, alternating)
  double *C_out, *counts_out, *csq_out, *Bpack_out;
  int disp_in, pad2_;         // what `slot` measures: 1 = against the tables this step read (the next step's labels were taken
                              // one update ago), 0 = against what C_out held before (two updates ago: the pipeline's steady state)
  SuspEntry *tab;             // [AHEAD_TAB_CAP] rows the bound cannot decide, with their candidate centroids
  int32_t *cnt;               // [4]: entries appended to `tab`; flags of rows the block cannot serve (1 table full, 2 open row at
                              //      position 0, 4 not finite, 8 list of open rows full); open rows appended to `hl`
  // open rows: more than three candidates.  Their distance to EVERY centroid is evaluated exactly, one step ahead: the wave
  // that has just updated centroid j (the M-step of the stratum's previous step in the block) computes the contract's d(x, c_j)
  // for the next step's open rows; the next step's waves take the first minimum over j.
  int32_t *hl;                // [hcap] this step's open rows
  double *hd;                 // [hcap][kst] their distances, written by the previous step's M-step
  double *td;                 // [AHEAD_TAB_CAP][3] the distances of the table's candidates, written by the previous step's M-step
  const int32_t *nx_cnt;      // the stratum's NEXT step in the block (null at the block's last position): its counters,
  const int32_t *nx_hl;       //   its open rows,
  const double *nx_Xg;        //   its batch copy,
  double *nx_hd;              //   where the open rows' distances go,
  const SuspEntry *nx_tab;    //   its table
  double *nx_td;              //   and where its candidates' distances go
  int hcap, kst;              // most open rows per step (by the row length: they wait in LDS); row stride of hd
};
struct AheadPack {
  AheadDev a[4];
};
#define AHEAD_TAB_CAP 256
#define AHEAD_SLOTS 64
#define HAMSM_STOP_AHEAD 100  // internal: a look-ahead block was given up at this step; run_steps relaunches from it itself
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

// A large call's E-step runs through the assignment kernels (hamsm_assign_rows: ranking pass with a proven bound + exact f64
// kernel for what it cannot certify -- the labels are the contract's); this kernel turns its output, labels per
// FRAME of the call's span, into what the M-step reads: membership bitmaps by batch row, the batch rows as f64, the labels by
// batch row.  Workgroup 0 of each stratum evaluates the tests that ride on k_step_assign's workgroup 0.
struct RowsPack {
  int off[PACK_NY];  // label offset of the step's stratum (labels of the span carry it)
};
template <typename XT>
__global__ __launch_bounds__(256) void k_rows_to_bits(const StepPack pk, const RowsPack rp, const int32_t *__restrict__ span_labels,
                                                      int64_t ldx, int D, int KS, double ratio) {
  const StepDev sdesc = pk.s[blockIdx.y];
  const StepG st(sdesc);
  const long long r0 = (long long)blockIdx.x * 64;
  if (r0 >= st.n_b) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  __shared__ double sd4[4];
  if (blockIdx.x == 0 && (st.flags & 6) && st.idx < *gp(sdesc.halt)) {
    // (k_step_assign's block: `counts < ratio * max(counts)` for the stratum's previous step, `(counts == 0).any()` before this one)
    const double inf = __builtin_huge_val();
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
      if (lane == 0) sd4[wv] = mx;
      __syncthreads();
      mx = sd4[0];
#pragma unroll
      for (int w = 1; w < 4; w++) mx = sd4[w] > mx ? sd4[w] : mx;
      const double thr = ratio * mx;
      int low = 0;
      for (int j = tid; j < st.K; j += 256) low += st.counts[j] < thr;
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
  // four threads per batch row
  const long long r = r0 + (tid >> 2);
  const int part = tid & 3;
  if (r >= st.n_b) return;
  const long long f = st.rows ? (long long)st.rows[r] : r;
  const GAS XT *X = (const GAS XT *)st.X;
  const int ldg = 4 * KS;
  for (int k = part; k < ldg; k += 4) st.Xg[(size_t)r * ldg + k] = k < D ? (double)X[f * ldx + k] : 0.0;
  if (part == 0) {
    int j = ((const GAS int32_t *)gp(span_labels))[f] - rp.off[blockIdx.y];
    j = (j >= 0 && j < st.K) ? j : 0;
    st.labels[r] = j;
    __hip_atomic_fetch_or(&st.mbits[(size_t)j * sdesc.wstride + (r >> 6)], 1ull << (r & 63), __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT);
  }
}

// M-step: one wave per centroid.  Round trip 1 fetches the centroid's membership bitmap (one word per lane), its count
// and its row; the set bits ARE the members in ascending sample order (no list to sort, no cap on the cluster size):
// each lane expands its word into the wave's LDS strip at the position given by a prefix sum of the popcounts.
// Round trip 2 fetches up to 64 member rows at once (lane = feature); they are added in sample order
// (_k_means_minibatch.pyx:59-108).
#define UPD_LIST 1024
#define UPD_SC 768  // member rows the strip holds per pass: ints [0, 768); weights of a 64-member chunk: ints [768, 896)
template <typename XT, int H, bool BYVAL, int UPD_NBK>
__global__ __launch_bounds__(256) void k_step_update(const StepDev *__restrict__ steps, const StepPack pk, int64_t ldx, int D,
                                                     int KS) {
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
  const int j = blockIdx.x * 4 + wv;
  if (j >= st.K) return;
  const int nw = (int)((st.n_b + 63) >> 6);
  GAS unsigned long long *mb = st.mbits + (size_t)j * sdesc.wstride;
  const int hv = *gp(sdesc.halt);
  const unsigned long long w0 = mb[lane];  // (wstride is a multiple of 64: always inside the centroid's row)
  const double cnt0 = st.counts[j];
  double c0[H];
#pragma unroll
  for (int h = 0; h < H; h++) {
    const int k = lane + 64 * h;
    c0[h] = st.C[(size_t)j * D + (k < D ? k : D - 1)];
  }
  SPROF_W(1, 1)
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
        int inc = pc;  // inclusive prefix sum over the lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int t = __shfl_up(inc, off, 64);
          inc += lane >= off ? t : 0;
        }
        excl[bk] = inc - pc;
        base[bk] = blk;
        blk += __shfl(inc, 63, 64);
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
      for (int m0 = 0; m0 < np; m0 += 64 / H) {
        const int cm = np - m0 < 64 / H ? np - m0 : 64 / H;
        if (weighted) {
          if (lane < cm) {
            const long long r = srow[m0 + lane];
            swl[lane] = st.wf[st.rows ? (long long)st.rows[r] : r];
          }
          for (int q = 0; q < cm; q++) wsum = wsum + swl[q];  // sample order (pyx:80-85)
        }
        if (cm <= 16 / H) upd_add_listed<16 / H, H>(acc, Xg, ldg, srow + m0, swl, weighted, 0, cm, D, lane);
        else if (cm <= 32 / H) upd_add_listed<32 / H, H>(acc, Xg, ldg, srow + m0, swl, weighted, 0, cm, D, lane);
        else upd_add_listed<64 / H, H>(acc, Xg, ldg, srow + m0, swl, weighted, 0, cm, D, lane);
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
  SPROF_W(1, 5)
#if defined(STEP_PROFILE) && STEP_PROFILE < 2
  SPROF(1, 5)
#endif
  SPROF_FLUSH(1)
}

// ---- pipelined training: the E-step two iterations ahead (an experiment, off by default) --------------------------
// The chain E(t) -> M(t) -> E(t+1) -> ... is what bounds a training pass: two dependent launches per WE iteration.  But the
// centroids of a stratum move less and less (by ~1/count per step), and a label is an argmin: it survives any displacement that is
// small against the gap between a row's nearest and second-nearest centroid.  So ONE launch per iteration (k_step_pipe) holds the
// M-step of iteration i and, beside it, the E-step of iteration i + 2 against the centroid tables that M-step READS (it writes
// the stratum's other set of tables: two sets, alternating) -- labels are two updates old when they are used.
//
// Certificate.  Write S = |x|, R_j = |x - c_j| for the centroids the E-step saw, and let every centroid have moved by at most B
// (Euclidean) when the labels are used.  The contract's distances are d_j = csq_j - 2 x.c_j (f64 fma chains), i.e.
// R_j^2 - S^2 up to a rounding error of at most g (2S + R_j)^2, g = (D + 6) 2^-53.  Centroid j cannot be the argmin then if
//     (R_j - B)^2 - g' (..)  >  (R_1 + B)^2 + g' (..)   <=   R_j^2 - R_1^2 > 2 B (R_1 + R_j) + k [(2S + R_j + B)^2 + (2S + R_1 + B)^2],
// with R_j > 2B, where 1 is the nearest centroid the E-step saw and k = 2^-38 covers every rounding above, and the 2^-40 by which
// the ranked values are perturbed (below), many times over.  The left side grows faster in R_j than the right once R_j > 2B, so
// the centroids are tested in order of distance: the second-nearest passes -> the label is FINAL and the row joins its
// centroid's membership bitmap as in the step-by-step E-step; else the first that passes ends the row's list of CANDIDATES (2 or
// 3 of them: the row is undecided and goes to the step's table); a fourth candidate makes the row OPEN (its distance to every
// centroid will be needed); a row or a distance that is not finite, a full table or a full list of open rows make the step
// unservable (`cnt[1]`).  Exact ties of the step-by-step rule (first minimum wins) are undecided rows here.
// Ranking: the sweep keeps each lane's four smallest distances in order with a min / max network (7 instructions per
// distance); the centroid's index rides in the 12 low mantissa bits of the value (a relative change below 2^-40, charged to k).
//
// The bound B is a GUESS made when the E-step is launched: alpha x the displacement the latest finished M-step measured.  The
// M-step that uses the labels checks it against the displacement really seen (`slot`: the M-step before it took the largest
// |c_new - c_as_the_E-step_saw_it| over all centroids -- the other set of tables holds exactly those old rows) and checks
// `cnt[1]`.  If either fails the stratum's chain stops at that step with HAMSM_STOP_AHEAD, nothing of it applied, and run_steps
// relaunches from there (the tables brought home first), the first two steps of the new segment with exact labels.
#define AHEAD_KAPPA 3.637978807091713e-12   // 2^-38
#define AHEAD_EPS 9.5367431640625e-07       // 2^-20
#define AHEAD_BIG 1e300                     // "no such centroid" (finite: the index bits must not turn an infinity into a NaN)
#define AHEAD_STRIP 1024                    // ints of LDS per wave of the M-step (4 KB: the member list and its weights, then the new row)
#define AHEAD_HX 512                        // doubles of LDS per wave for the next step's rows (34 KB per workgroup in all: four per CU)
#define AHEAD_HCAP 32                       // most open rows per step (fewer for long rows: they wait in LDS)
#define AHEAD_KMAX 4096                     // centroids per stratum the 12 index bits can name

__device__ __forceinline__ double f64_min(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double f64_max(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// k1 <= k2 <= k3 <= k4 keep the four smallest keys seen
__device__ __forceinline__ void top4_insert(double k, double &k1, double &k2, double &k3, double &k4) {
  double t;
  t = f64_min(k, k1), k = f64_max(k, k1), k1 = t;
  t = f64_min(k, k2), k = f64_max(k, k2), k2 = t;
  t = f64_min(k, k3), k = f64_max(k, k3), k3 = t;
  k4 = f64_min(k, k4);
}
__device__ __forceinline__ double key_of(double d, int j) {
  return __hiloint2double(__double2hiint(d), (int)(((unsigned)__double2loint(d) & 0xFFFFF000u) | (unsigned)j));
}
__device__ __forceinline__ int key_index(double k) { return __double2loint(k) & 0xFFF; }

template <typename XT, int KS, int JB, int RT>
__device__ __forceinline__ void ahead_assign(const StepDev *__restrict__ steps, const AheadDev *__restrict__ aheads, const int yi,
                                             const unsigned bx, int64_t ldx, int D) {
  // RT tiles of 16 batch rows per workgroup: every centroid fragment a wave pulls through L2 serves RT MFMAs (this E-step
  // is off the pass's critical chain and bound by that traffic, unlike k_step_assign)
  constexpr int RW = 16 * RT;
  const StepDev sdesc = load_step_mem(steps, (unsigned)yi);
  const GAS AheadDev *ahp = gp(aheads + yi);
  const StepG st(sdesc);
  const long long r0 = (long long)bx * RW;
  if (r0 >= st.n_b) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const GAS XT *X = (const GAS XT *)st.X;
  // the bound of this step (every workgroup derives the same value; workgroup 0 records it for the M-step)
  const double *ref_slot = ahp->ref_slot;
  const double ref_scale = ahp->ref_scale;
  double B = 0.0;
  if (ref_slot) {
    double m = gp(ref_slot)[lane & (AHEAD_SLOTS - 1)];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double o = __shfl_xor(m, off, 64);
      m = (o > m || o != o) ? o : m;  // (a NaN displacement stays a NaN: the bound fails its check)
    }
    B = ref_scale * m;
  }
  long long f[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    const long long r = r0 + rt * 16 + (lane & 15);
    const long long rc = r < st.n_b ? r : st.n_b - 1;
    f[rt] = st.rows ? (long long)st.rows[rc] : rc;
  }
  const int njt = (st.K + 15) >> 4;
  const double inf = __builtin_huge_val();
  double b[JB][KS], cs[JB][4];
  auto load_group = [&](int t0) {
#pragma unroll
    for (int u = 0; u < JB; u++) {
      const int jt = t0 + 4 * u < njt ? t0 + 4 * u : njt - 1;
#pragma unroll
      for (int q = 0; q < 4; q++) cs[u][q] = st.csq[jt * 16 + (lane >> 4) + 4 * q];
#pragma unroll
      for (int ks = 0; ks < KS; ks++) b[u][ks] = st.Bpack[((size_t)jt * KS + ks) * 64 + lane];
    }
  };
  load_group(wv);
  double a[RT][KS];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int k = ks * 4 + (lane >> 4);
      a[rt][ks] = (double)X[f[rt] * ldx + (k < D ? k : D - 1)];
    }
  __shared__ double sk[4][4][RW];
  __shared__ int snan[4][RT];
  GAS int32_t *cnt = gp(ahp->cnt);
  // the batch rows as f64 for the M-step: row tile rt is written by wave (rt + 1) & 3 (as in the step-by-step E-step)
  auto write_rows = [&]() {
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
      if (((rt + 1) & 3) == wv) {
        const long long r = r0 + rt * 16 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) st.Xg[(size_t)r * (4 * KS) + ks * 4 + (lane >> 4)] = a[rt][ks];
      }
  };
  if (ahp->pos == 0) {
    // the segment's first steps: these ARE the centroids their M-step will see -- the step-by-step E-step's rule, exactly
    double bd[RT];
    int bi[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
      bd[rt] = inf;
      bi[rt] = 0x7fffffff;
    }
    for (int t0 = wv; t0 < njt; t0 += 4 * JB) {
      if (t0 != wv) load_group(t0);
#pragma unroll
      for (int u = 0; u < JB; u++) {
        const int jt = t0 + 4 * u;
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
              const double d = __builtin_fma(-2.0, acc[rt][q], cs[u][q]);
              const bool lt = d < bd[rt];
              bd[rt] = lt ? d : bd[rt];
              bi[rt] = lt ? j0 + 4 * q : bi[rt];
            }
        }
      }
    }
    write_rows();
    double *sdv = &sk[0][0][0];       // [4][RW] as one array (the ranking arrays are free on this path)
    int *sji = (int *)&sk[1][0][0];   // [4][RW]
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
      double d = bd[rt];
      int j = bi[rt];
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const double d2 = __shfl_xor(d, off, 64);
        const int j2 = __shfl_xor(j, off, 64);
        const bool take = (d2 < d) || (d2 == d && j2 < j);
        d = take ? d2 : d;
        j = take ? j2 : j;
      }
      if (lane < 16) {
        sdv[wv * RW + rt * 16 + lane] = d;
        sji[wv * RW + rt * 16 + lane] = j;
      }
    }
    __syncthreads();
    if (bx == 0 && tid == 0) *gp(ahp->bound) = 0.0;
    if (tid < RW && r0 + tid < st.n_b) {
      double d = sdv[tid];
      int j = sji[tid];
#pragma unroll
      for (int w = 1; w < 4; w++) {
        const double d2 = sdv[w * RW + tid];
        const int j2 = sji[w * RW + tid];
        const bool take = (d2 < d) || (d2 == d && j2 < j);
        d = take ? d2 : d;
        j = take ? j2 : j;
      }
      if (j == 0x7fffffff) j = 0;  // all-NaN row: label 0 like the oracle
      const long long rr = r0 + tid;
      __hip_atomic_fetch_or(&st.mbits[(size_t)j * sdesc.wstride + (rr >> 6)], 1ull << (rr & 63), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  double kk[RT][4];
  bool odd[RT];  // a distance that is not a number
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    kk[rt][0] = kk[rt][1] = kk[rt][2] = kk[rt][3] = AHEAD_BIG;
    odd[rt] = false;
  }
  for (int t0 = wv; t0 < njt; t0 += 4 * JB) {
    if (t0 != wv) load_group(t0);
#pragma unroll
    for (int u = 0; u < JB; u++) {
      const int jt = t0 + 4 * u;
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
            // (rows of the last tile past K carry csq = +inf: they rank as "no such centroid")
            const double d = __builtin_fma(-2.0, acc[rt][q], f64_min(cs[u][q], AHEAD_BIG));
            odd[rt] = odd[rt] || (d != d);
            top4_insert(key_of(d, j0 + 4 * q), kk[rt][0], kk[rt][1], kk[rt][2], kk[rt][3]);
          }
      }
    }
  }
  write_rows();
  double x2r[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    // |x|^2 of the lane's row (features past D were read at the clamped index: not counted)
    double x2 = 0.0;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int k = ks * 4 + (lane >> 4);
      x2 = k < D ? __builtin_fma(a[rt][ks], a[rt][ks], x2) : x2;
    }
    x2 += __shfl_xor(x2, 16, 64);
    x2 += __shfl_xor(x2, 32, 64);
    x2r[rt] = x2;
    // lanes l, l^16, l^32, l^48 hold the same batch row: merge their four best
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const double e1 = __shfl_xor(kk[rt][0], off, 64), e2 = __shfl_xor(kk[rt][1], off, 64), e3 = __shfl_xor(kk[rt][2], off, 64),
                   e4 = __shfl_xor(kk[rt][3], off, 64);
      top4_insert(e1, kk[rt][0], kk[rt][1], kk[rt][2], kk[rt][3]);
      top4_insert(e2, kk[rt][0], kk[rt][1], kk[rt][2], kk[rt][3]);
      top4_insert(e3, kk[rt][0], kk[rt][1], kk[rt][2], kk[rt][3]);
      top4_insert(e4, kk[rt][0], kk[rt][1], kk[rt][2], kk[rt][3]);
    }
    const unsigned long long oddm = __ballot(odd[rt]);
    if (lane < 16) {
#pragma unroll
      for (int q = 0; q < 4; q++) sk[wv][q][rt * 16 + lane] = kk[rt][q];
    }
    if (lane == 0) snan[wv][rt] = (int)((oddm | (oddm >> 16) | (oddm >> 32) | (oddm >> 48)) & 0xFFFF);  // by batch row
  }
  // (wave 0's lanes hold x2 of rows (rt, lane & 15): thread t < RW needs that of row t)
  __shared__ double sx2[RW];
  if (wv == 0 && lane < 16) {
#pragma unroll
    for (int rt = 0; rt < RT; rt++) sx2[rt * 16 + lane] = x2r[rt];
  }
  __syncthreads();
  if (bx == 0 && tid == 0) *gp(ahp->bound) = B;
  if (tid < RW && r0 + tid < st.n_b) {
    double k1 = sk[0][0][tid], k2 = sk[0][1][tid], k3 = sk[0][2][tid], k4 = sk[0][3][tid];
#pragma unroll
    for (int w = 1; w < 4; w++) {
#pragma unroll
      for (int q = 0; q < 4; q++) top4_insert(sk[w][q][tid], k1, k2, k3, k4);
    }
    const int rt_ = tid >> 4;
    const bool row_odd = ((snan[0][rt_] | snan[1][rt_] | snan[2][rt_] | snan[3][rt_]) >> (tid & 15)) & 1;
    const long long rr = r0 + tid;
    const double x2 = sx2[tid];
    const double S = __builtin_sqrt(x2);
    const double D1 = k1 + x2, R1 = __builtin_sqrt(D1 > 0.0 ? D1 : 0.0);
    // true when the centroid ranked with key kj cannot become the argmin under any displacement <= B of every centroid
    auto out_of_reach = [&](double kj) {
      if (kj >= 1e299) return true;  // no such centroid
      const double Dj = kj + x2, Rj = __builtin_sqrt(Dj > 0.0 ? Dj : 0.0);
      const double u = 2.0 * S + Rj + B, v = 2.0 * S + R1 + B;
      const double need = 2.0 * B * (R1 + Rj) * (1.0 + AHEAD_EPS) + AHEAD_KAPPA * (u * u + v * v);
      return (Rj > 2.0 * B * (1.0 + AHEAD_EPS) + AHEAD_EPS * (S + Rj)) && (Dj - D1 > need);
    };
    const bool finite = !row_odd && (x2 < inf) && (k1 > -1e299) && (k1 < 1e299) && (B < inf);
    const int j1 = key_index(k1), j2 = key_index(k2), j3 = key_index(k3);
    if (finite && out_of_reach(k2)) {
      // (result unused: a non-returning atomic)
      __hip_atomic_fetch_or(&st.mbits[(size_t)j1 * sdesc.wstride + (rr >> 6)], 1ull << (rr & 63), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
    } else if (finite && (out_of_reach(k3) || out_of_reach(k4))) {
      const bool three = !out_of_reach(k3);
      const int at = __hip_atomic_fetch_add(&cnt[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (at < AHEAD_TAB_CAP) {
        GAS int *e = (GAS int *)gp(ahp->tab) + 4 * at;
        e[0] = (int)rr;
        e[1] = j1;
        e[2] = j2;
        e[3] = three ? j3 : -1;
      } else {
        __hip_atomic_fetch_or(&cnt[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (finite) {
      // an open row (a fourth candidate): every centroid's exact distance, by the M-step of the stratum's previous step
      const int at = __hip_atomic_fetch_add(&cnt[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (at < ahp->hcap) gp(ahp->hl)[at] = (int)rr;
      else __hip_atomic_fetch_or(&cnt[1], 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __hip_atomic_fetch_or(&cnt[1], 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <typename XT, int KS, int JB, int RT>
__global__ __launch_bounds__(256) void k_step_assign_ahead(const StepDev *__restrict__ steps, const AheadDev *__restrict__ aheads,
                                                           const int *__restrict__ ylist, int64_t ldx, int D) {
  ahead_assign<XT, KS, JB, RT>(steps, aheads, ((const GAS int *)ylist)[blockIdx.y], blockIdx.x, ldx, D);
}

// M-step of a look-ahead block: k_step_update plus (i) the check of the block's bound, (ii) the low-count / zero-count tests
// that ride on the E-step in the step-by-step path (here every workgroup evaluates them, they all see the same counts),
// (iii) the rows the E-step left undecided or open, by the distances the previous step's waves left behind, (iv) the
// displacement bookkeeping, (v) this centroid's distances to the NEXT step's open rows and to the rows of the next step's
// table that name it.  One wave per centroid, descriptors in the kernel arguments.
#define AHEAD_ITEMS 64  // rows of the next step a wave evaluates in one round (lane = row)
// AHEAD_PROFILE (tools/aheadprof.py): wall-clock stamps of the M part's phases, wave 0 of every workgroup of stratum row 0
#ifdef AHEAD_PROFILE
#define APF_N 8
__device__ unsigned long long g_ahead_prof[64][APF_N];
extern "C" int hamsm_debug_ahead_profile(unsigned long long *h_out) {
  return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_ahead_prof), sizeof(unsigned long long) * 64 * APF_N) == hipSuccess ? 0 : -1;
}
#define APF(slot)                                                                       \
  {                                                                                     \
    __builtin_amdgcn_s_waitcnt(0);                                                      \
    if (blockIdx.y == 0 && bx < 64 && threadIdx.x == 0) g_ahead_prof[bx][slot] = wall_clock64(); \
  }
#else
#define APF(slot)
#endif
template <typename XT, int H>
__device__ __forceinline__ void ahead_update(const StepDev &sdesc, const AheadDev &ah, const unsigned bx, int64_t ldx, int D, int KS,
                                             double ratio, int *lds_i) {
  const StepG st(sdesc);
  if ((int)bx * 4 >= st.K) return;
  APF(0)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int *list = lds_i + wv * AHEAD_STRIP;
  unsigned *srow = (unsigned *)list;
  double *swl = (double *)(list + UPD_SC);
  const GAS double *Xg = st.Xg;
  const int ldg = 4 * KS;
  const int j = bx * 4 + wv;
  const bool active = j < st.K;
  const int jc = active ? j : st.K - 1;  // (idle waves of the last workgroup load what the last centroid's wave loads; they store nothing)
  const double inf = __builtin_huge_val();
  GAS unsigned long long *mb = st.mbits + (size_t)jc * sdesc.wstride;
  // round trip 1: everything whose address is known from the descriptor
  const int hv = *gp(sdesc.halt);
  const unsigned long long w0 = mb[lane];
  const double cnt0 = st.counts[jc];
  const double csq0 = st.csq[jc];
  GAS double *C_out = gp(ah.C_out);
  double c0[H], sn[H];
#pragma unroll
  for (int h = 0; h < H; h++) {
    const int k = lane + 64 * h;
    c0[h] = st.C[(size_t)jc * D + (k < D ? k : D - 1)];
    sn[h] = C_out[(size_t)jc * D + (k < D ? k : D - 1)];  // this centroid two updates ago (steady state)
  }
  if (ah.disp_in) {
#pragma unroll
    for (int h = 0; h < H; h++) sn[h] = c0[h];
  }
  double seen = 0.0;
  if (ah.prev_slot) {
    seen = gp(ah.prev_slot)[lane & (AHEAD_SLOTS - 1)];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double o = __shfl_xor(seen, off, 64);
      seen = (o > seen || o != o) ? o : seen;
    }
  }
  const double bound = *gp((const double *)ah.bound);
  const GAS int32_t *cnt = gp((const int32_t *)ah.cnt);
  const int n_und = cnt[0], unserved = cnt[1];
  int n_open = cnt[2];
  n_open = n_open < ah.hcap ? n_open : ah.hcap;
  const int r_open = gp((const int32_t *)ah.hl)[lane & (AHEAD_HCAP - 1)];
  // this step's table: an entry per lane, with the candidates' distances the previous step's waves left behind
  const GAS v4i32 *tab = (const GAS v4i32 *)gp(ah.tab);
  const GAS double *td = gp((const double *)ah.td);
  v4i32 ent = tab[lane];  // (entries past n_und are never looked at)
  double td0 = td[3 * lane], td1 = td[3 * lane + 1], td2 = td[3 * lane + 2];
  // the next step's rows this wave owes a distance to: its open rows, then the entries of its table that name this centroid.
  // Lists now, rows into LDS behind them: both round trips overlap with this step's own
  int n_nx = 0, n_nt = 0;
  if (ah.nx_cnt) {
    n_nt = gp(ah.nx_cnt)[0];
    n_nx = gp(ah.nx_cnt)[2];
  }
  n_nx = n_nx < ah.hcap ? n_nx : ah.hcap;
  n_nt = n_nt < AHEAD_TAB_CAP ? n_nt : AHEAD_TAB_CAP;
  const int DPH = (D + 1) & ~1;
  const int cap = AHEAD_HX / DPH;  // rows the LDS holds per round (>= hcap)
  double *hx = (double *)(lds_i + 4 * AHEAD_STRIP) + wv * AHEAD_HX;
  int *meta = lds_i + 4 * AHEAD_STRIP + 2 * 4 * AHEAD_HX + wv * (2 * AHEAD_ITEMS);  // per item: where its distance goes, its row
  const GAS double *nXg = gp(ah.nx_Xg);
  const GAS v4i32 *ntab = (const GAS v4i32 *)gp(ah.nx_tab);
  int n_items = 0;  // items of the first round: staged in LDS now, evaluated at the end
  // stage rows `cnt_` (row numbers by lane, in `rows_l`) at item slots [at, at + cnt_): all lanes load a row's features
  auto stage = [&](int rows_l, int cnt_, int at) {
    for (int h0 = 0; h0 < cnt_; h0 += 4) {
      double t[4][H];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int rq = __shfl(rows_l, h0 + q < cnt_ ? h0 + q : cnt_ - 1, 64);
#pragma unroll
        for (int h = 0; h < H; h++) {
          const int k = lane + 64 * h;
          t[q][h] = nXg[(size_t)rq * ldg + (k < D ? k : D - 1)];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (h0 + q < cnt_) {
#pragma unroll
          for (int h = 0; h < H; h++) {
            const int k = lane + 64 * h;
            if (k < D) hx[(at + h0 + q) * DPH + k] = t[q][h];
          }
        }
    }
  };
  if (active && n_nx > 0) {
    const int r_nx = gp(ah.nx_hl)[lane & (AHEAD_HCAP - 1)];
    stage(r_nx, n_nx, 0);
    if (lane < n_nx) {
      meta[2 * lane] = -1 - lane;  // an open row: its distance goes to nx_hd[lane][j]
      meta[2 * lane + 1] = 0;
    }
    n_items = n_nx;
  }
  int nt_done = 0;  // entries of the next table looked at so far
  if (active && n_nt > 0) {
    // the first 64 entries of the next step's table (the later ones, and what does not fit the LDS, wait for the end)
    const v4i32 e = ntab[lane];
    const int slot = (lane < n_nt) ? (e.y == j ? 0 : (e.z == j ? 1 : (e.w == j ? 2 : -1))) : -1;
    const unsigned long long mm = __ballot(slot >= 0);
    const int mine = __popcll(mm);
    const int room = (cap < AHEAD_ITEMS ? cap : AHEAD_ITEMS) - n_items;
    if (mine <= room) {
      const int rank = __popcll(mm & ((1ull << lane) - 1));
      // compact the rows of my entries to lanes 0 .. mine-1
      if (slot >= 0) {
        meta[2 * (n_items + rank)] = 3 * lane + slot;  // where the distance goes: nx_td[3 * entry + slot]
        meta[2 * (n_items + rank) + 1] = e.x;
      }
      const int rows_l = lane < mine ? meta[2 * (n_items + lane) + 1] : 0;
      if (mine) stage(rows_l, mine, n_items);
      n_items += mine;
      nt_done = n_nt < 64 ? n_nt : 64;
    }
  }
  // the stratum's counts for the tests: every WAVE reads all of them (K / 64 loads per lane) and decides for itself -- the
  // same counts, the same verdict everywhere, no barrier
  double cj_t = -inf;
  int z_t = 0;
  if (st.flags & 6)
    for (int q = lane; q < st.K; q += 64) {
      const double cq = st.counts[q];
      z_t |= (cq == 0.0);
      cj_t = cq > cj_t ? cq : cj_t;
    }
  if (st.idx >= hv) return;  // the stratum's chain stopped before this step
  APF(1)
  // (i) the premises of this step's labels
  if (!(seen <= bound) || unserved) {
    if (bx == 0 && tid == 0) {
      *gp(sdesc.halt) = st.idx;
      *gp(sdesc.reason) = HAMSM_STOP_AHEAD;
    }
    return;
  }
  // (ii) _kmeans.py:1645 for the stratum's previous step / :2037 before this one (k_step_assign's block, evaluated by every wave)
  if (st.flags & 6) {
    double mx = cj_t;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double o = __shfl_xor(mx, off, 64);
      mx = o > mx ? o : mx;
    }
    int any_low = 0;
    if (st.flags & 4) {
      const double thr = ratio * mx;
      int low = 0;
      for (int q = lane; q < st.K; q += 64) low += st.counts[q] < thr;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) low += __shfl_xor(low, off, 64);
      if (low && bx == 0 && wv == 0 && lane == 0)
        __hip_atomic_fetch_add((GAS int32_t *)gp(sdesc.prev_report) + 1, low, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      any_low = low;
    }
    const int any_z = (st.flags & 2) ? (__ballot(z_t != 0) != 0ull) : 0;
    if (any_low || any_z) {
      if (bx == 0 && tid == 0) {
        if (any_low) {
          *gp(sdesc.halt) = st.idx;
          *gp(sdesc.reason) = HAMSM_STOP_LOW;
        } else {
          st.report[0] = 1;
          *gp(sdesc.halt) = st.idx;
          *gp(sdesc.reason) = HAMSM_STOP_ZERO;
        }
      }
      return;
    }
  }
  GAS double *slot = gp(ah.slot);
  if (!active) return;
  unsigned long long w = w0;
  APF(2)
  // (iii-a) open rows: first minimum over all centroids of the distances the previous step's waves left behind.  Four rows'
  // values are requested together (one round trip per four rows), then reduced
  for (int h0 = 0; h0 < n_open; h0 += 4) {
    double bd[4];
    int bj[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) {
      bd[q4] = inf;
      bj[q4] = 0x7fffffff;
    }
    for (int q = lane; q < st.K; q += 64) {  // (a lane's centroids ascend: strict < keeps its lowest index)
      double dq[4];
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++) {
        const int h = h0 + q4 < n_open ? h0 + q4 : n_open - 1;
        dq[q4] = gp((const double *)ah.hd)[(size_t)h * ah.kst + q];
      }
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++) {
        const bool lt = dq[q4] < bd[q4];
        bd[q4] = lt ? dq[q4] : bd[q4];
        bj[q4] = lt ? q : bj[q4];
      }
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) {
      double d = bd[q4];
      int jm = bj[q4];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double d2 = __shfl_xor(d, off, 64);
        const int j2 = __shfl_xor(jm, off, 64);
        const bool take = (d2 < d) || (d2 == d && j2 < jm);
        d = take ? d2 : d;
        jm = take ? j2 : jm;
      }
      if (jm == 0x7fffffff) jm = 0;  // (no distance is a number: label 0 like the step-by-step rule)
      if (h0 + q4 < n_open) {
        const int rh = __shfl(r_open, h0 + q4, 64);
        if (jm == j && lane == (rh >> 6)) w |= 1ull << (rh & 63);
      }
    }
  }
  // (iii-b) undecided rows: the first minimum among the entry's candidates, by the distances left behind
  {
    const int nt = n_und < AHEAD_TAB_CAP ? n_und : AHEAD_TAB_CAP;
    for (int base = 0; base < nt; base += 64) {
      if (base) {
        ent = tab[base + lane];
        td0 = td[3 * (base + lane)], td1 = td[3 * (base + lane) + 1], td2 = td[3 * (base + lane) + 2];
      }
      // ascending centroid index is not the entry's order: strictly smaller, or equal with the lower index, replaces
      double bd = td0;
      int bj = ent.y;
      bool take = (td1 < bd) || (td1 == bd && ent.z < bj) || (bd != bd && td1 == td1);
      bd = take ? td1 : bd;
      bj = take ? ent.z : bj;
      take = ent.w >= 0 && ((td2 < bd) || (td2 == bd && ent.w < bj) || (bd != bd && td2 == td2));
      bd = take ? td2 : bd;
      bj = take ? ent.w : bj;
      unsigned long long mm = __ballot(base + lane < nt && bj == j);
      while (mm) {
        const int src = __builtin_ctzll(mm);
        mm &= mm - 1;
        const int rr = __shfl(ent.x, src, 64);
        if (lane == (rr >> 6)) w |= 1ull << (rr & 63);
      }
    }
  }
  // (the rest is k_step_update; batches of a look-ahead block have at most 4096 rows: one block of bitmap words)
  if (w0) mb[lane] = 0ull;
  APF(3)
  const bool weighted = st.wf != nullptr;
  double acc[H];
#pragma unroll
  for (int h = 0; h < H; h++) acc[h] = c0[h] * cnt0;
  double wsum = 0.0;
  const int pc = __popcll(w);
  int inc = pc;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    inc += lane >= off ? t : 0;
  }
  const int excl = inc - pc, blk = __shfl(inc, 63, 64);
  APF(5)
  for (int p0 = 0; p0 < blk; p0 += UPD_SC) {
    {
      unsigned long long t = w;
      int idx = excl - p0;
      while (t) {
        const int bit = __builtin_ctzll(t);
        t &= t - 1;
        if (idx >= 0 && idx < UPD_SC) srow[idx] = (unsigned)(lane * 64 + bit);
        idx++;
      }
    }
    APF(6)
    const int np = blk - p0 < UPD_SC ? blk - p0 : UPD_SC;
    for (int m0 = 0; m0 < np; m0 += 16 / H) {  // (16 / H rows in flight: a quarter of k_step_update's, for the registers of the fused launch)
      const int cm = np - m0 < 16 / H ? np - m0 : 16 / H;
      if (weighted) {
        if (lane < cm) {
          const long long r = srow[m0 + lane];
          swl[lane] = st.wf[st.rows ? (long long)st.rows[r] : r];
        }
        for (int q = 0; q < cm; q++) wsum = wsum + swl[q];
      }
      upd_add_listed<16 / H, H>(acc, Xg, ldg, srow + m0, swl, weighted, 0, cm, D, lane);
    }
  }
  if (!weighted) wsum = (double)blk;
  APF(4)
#ifdef AHEAD_PROFILE
  if (blockIdx.y == 0 && bx < 64 && threadIdx.x == 0) g_ahead_prof[bx][7] = (unsigned long long)blk | ((unsigned long long)n_open << 16) | ((unsigned long long)n_und << 32);
#endif
  const bool moved = blk > 0 && wsum > 0;  // (no member, or no weight: centre unchanged, pyx:87 / :105-108)
  double *vb = (double *)list;
  double a = csq0;
  {
    const double cnt1 = moved ? cnt0 + wsum : cnt0;
    const double alpha = 1.0 / cnt1;
    double v[H], dsq = 0.0;
    GAS double *Bp_out = gp(ah.Bpack_out);
#pragma unroll
    for (int h = 0; h < H; h++) {
      const int k = lane + 64 * h;
      v[h] = moved ? acc[h] * alpha : c0[h];
      if (k < D) {
        C_out[(size_t)j * D + k] = v[h];
        Bp_out[((size_t)(j >> 4) * KS + (k >> 2)) * 64 + (k & 3) * 16 + (j & 15)] = v[h];
        const double t = v[h] - sn[h];
        dsq += t * t;
      }
    }
    if (lane == 0) gp(ah.counts_out)[j] = cnt1;
    // (iv) how far this centroid now is from where the next step's labels were taken (any summation order: the factor covers
    // it; NaN and inf stay what they are and fail the next step's check)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dsq += __shfl_xor(dsq, off, 64);
    if (lane == 0 && dsq != 0.0) {
      const double disp = __builtin_sqrt(dsq) * (1.0 + AHEAD_EPS);
      __hip_atomic_fetch_max((GAS unsigned long long *)slot + (j & (AHEAD_SLOTS - 1)), (unsigned long long)__double_as_longlong(disp),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int h = 0; h < H; h++) vb[lane + 64 * h] = v[h];
    if (moved) {
      a = 0.0;
      for (int k = 0; k + 1 < D; k += 2) {
        const double2 t = *(const double2 *)&vb[k];
        a = __builtin_fma(t.x, t.x, a);
        a = __builtin_fma(t.y, t.y, a);
      }
      if (D & 1) a = __builtin_fma(vb[D - 1], vb[D - 1], a);
    }
    if (lane == 0) gp(ah.csq_out)[j] = a;
  }
  // (v) what the next step needs of this centroid as it now stands: lane = row, the contract's chain over ascending k --
  // dot = fma(c_k, x_k, dot) from +0.0, what the matrix unit accumulates for this (row, centroid) pair in k_step_assign --
  // then d = fma(-2, dot, csq)
  auto evaluate = [&](int n_it) {
    const int it = lane < n_it ? lane : 0;
    const double *xs = hx + it * DPH;
    double dot = 0.0;
    for (int k = 0; k < D; k++) dot = __builtin_fma(vb[k], xs[k], dot);
    const double dl = __builtin_fma(-2.0, dot, a);
    if (lane < n_it) {
      const int m = meta[2 * lane];
      if (m < 0) gp(ah.nx_hd)[(size_t)(-1 - m) * ah.kst + j] = dl;
      else gp(ah.nx_td)[m] = dl;
    }
  };
  if (n_items) evaluate(n_items);
  // what did not fit the first round: the rest of the next step's table, `cap` rows at a time
  for (int base = nt_done; base < n_nt; base += 64) {
    const v4i32 e = ntab[base + lane];
    const int slot = (base + lane < n_nt) ? (e.y == j ? 0 : (e.z == j ? 1 : (e.w == j ? 2 : -1))) : -1;
    unsigned long long mm = __ballot(slot >= 0);
    while (mm) {
      // the first `cap` of the entries still open
      unsigned long long take = 0;
      int got = 0;
      unsigned long long t = mm;
      while (t && got < cap && got < AHEAD_ITEMS) {
        take |= t & (~t + 1);
        t &= t - 1;
        got++;
      }
      mm &= ~take;
      const bool in = (take >> lane) & 1;
      const int rank = __popcll(take & ((1ull << lane) - 1));
      if (in) {
        meta[2 * rank] = 3 * (base + lane) + slot;
        meta[2 * rank + 1] = e.x;
      }
      const int rows_l = lane < got ? meta[2 * lane + 1] : 0;
      stage(rows_l, got, 0);
      evaluate(got);
    }
  }
}

// one launch of the pipeline: blockIdx.y < n_m -> the M-step of a step whose labels are complete (descriptors by value);
// the other rows of the grid -> E-steps of later steps (descriptors in memory), against tables no M-step of this launch writes
template <typename XT, int KS, int JB, int H, int RT>
__global__ __launch_bounds__(256, 3) void k_step_pipe(const StepPack pk, const AheadPack ap, int n_m, const StepDev *__restrict__ steps,
                                                   const AheadDev *__restrict__ aheads, const int *__restrict__ ylist, int64_t ldx,
                                                   int D, double ratio) {
  extern __shared__ int lds_i[];
  if ((int)blockIdx.y < n_m) {
    // (copies: every field is fetched from the kernel-argument segment at once, not where it is first used)
    const StepDev sd_ = pk.s[blockIdx.y];
    const AheadDev ah_ = ap.a[blockIdx.y];
    ahead_update<XT, H>(sd_, ah_, blockIdx.x, ldx, D, KS, ratio, lds_i);
  } else {
    ahead_assign<XT, KS, JB, RT>(steps, aheads, ((const GAS int *)ylist)[blockIdx.y - n_m], blockIdx.x, ldx, D);
  }
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
__global__ __launch_bounds__(256) void k_step_reassign(const StepDev *__restrict__ steps, const StepPack pk, int64_t ldx, int D,
                                                       int KS, double ratio) {
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
    }
    base += total;
    __syncthreads();
  }
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

// ===================================================================================================
// Persistent training pass: every step of every stratum in ONE launch.
//
// A stratum's chain of partial_fits is a strict sequence E-step -> M-step -> E-step ...: all rows need all centroids, every
// centroid needs all its member rows.  As separate kernels each hand-off costs a kernel boundary plus the start and the
// drain of a grid (~4 us of each ~8 us kernel).  Here the workgroups that landed on one XCD serve one stratum for its
// whole chain: they share that XCD's L2, so the two hand-offs per step are an XCD-LOCAL barrier -- stores drained to L2
// (the vector L1 is write-through), one atomic on a counter, one polling lane, and every load of handed-off bytes a
// `global_load ... sc1` (bypasses the reader's L1, served by the shared L2).  No L2 write-back / invalidate (the
// agent-scope fences cost 1.7 us each) is needed, because no other XCD ever reads these bytes inside the launch.
//
// Placement is not assumed: a workgroup reads its XCD from the hardware register, registers there (rank within the XCD),
// one chip-wide arrival count closes the registration, and strata are dealt over the XCDs that actually received
// workgroups (an XCD serves several strata one chain after the other when there are more strata than populated XCDs).
// Every spin is bounded by the 100 MHz wall clock; a timeout sets `ctl->err` and the workgroup leaves (1 = during
// registration, nothing has been modified and the host takes the multi-kernel path; 2 = inside a chain: an error).
//
// Arithmetic, orders and the stop protocol are those of k_step_assign / k_step_update / k_step_reassign / k_step_check
// above (same descriptors, same report / halt / reason words): results are bit-identical to that path and to the oracle.
#ifndef PERSIST_ATOMIC_SCOPE
// Atomics on words shared by the workgroups of ONE XCD are performed in that XCD's L2 at workgroup scope (at agent scope
// they travel beyond the L2s: ~2x the latency).  -DPERSIST_ATOMIC_SCOPE=__HIP_MEMORY_SCOPE_AGENT builds the conservative form.
#define PERSIST_ATOMIC_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
struct PersistCtl {     // zeroed by one hipMemsetAsync before every launch
  unsigned arrive_all;  // registration: workgroups that have registered
  unsigned err;         // 0, 1 = registration timed out, 2 = a barrier timed out
  unsigned nwg_xcd[8];  // workgroups per XCD
  unsigned pad[6];
  unsigned bar[];       // [S] one monotonic arrival counter per stratum
};

template <typename T>
__device__ __forceinline__ T ld_l2(const GAS T *p) {  // global_load ... sc1: never served by this CU's L1
  if constexpr (sizeof(T) == 8) {
    unsigned long long v = __hip_atomic_load((const GAS unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(T, v);
  } else {
    static_assert(sizeof(T) == 4, "ld_l2: 4 or 8 bytes");
    unsigned v = __hip_atomic_load((const GAS unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(T, v);
  }
}

#define PERSIST_TMO_REG 2000000ull    // 20 ms of 100 MHz ticks: every workgroup of the grid resident and registered
#define PERSIST_TMO_BAR 30000000ull   // 300 ms per barrier (a phase takes microseconds)

// XCD-local barrier among the W workgroups that serve one stratum.  `sh_ok` is an LDS word of the caller.
__device__ __forceinline__ bool xcd_barrier(GAS unsigned *bar, unsigned target, GAS unsigned *err, int *sh_ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores and atomics have reached the L2
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, PERSIST_ATOMIC_SCOPE);
    int ok = 1;
    const unsigned long long t0 = wall_clock64();
    while ((int)(ld_l2(bar) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > PERSIST_TMO_BAR || ld_l2(err) != 0u) {
        __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    *sh_ok = ok;
  }
  __syncthreads();
  asm volatile("" ::: "memory");
  return *sh_ok != 0;
}

// The phases are separate (non-inlined) functions: each keeps only its own operands live (as one body the kernel needed more
// than 256 VGPRs).  LDS scratch is handed in by the kernel.
struct PersistLds {
  double sd[4][32];
  int sj[4][32];
  double shd[4];
  int shi[4];
  int lj[256], lr[256];
  int sh_i[16];
};

// low-count test of the stratum's previous step / zero-count test before this one -- workgroup 0 of k_step_assign
__device__ __attribute__((noinline)) void persist_precheck(const StepDev *sp, double ratio, PersistLds *L) {
  const StepDev &sdesc = *sp;
  const StepG st(sdesc);
  const int tid = threadIdx.x;
  const double inf = __builtin_huge_val();
  GAS int32_t *halt = gp(sdesc.halt);
  const GAS int32_t *prev = gp(sdesc.prev_report);
  int z = 0;
  double mx = -inf;
  for (int j = tid; j < st.K; j += 256) {
    double cj = ld_l2(st.counts + j);
    z |= (cj == 0.0);
    mx = cj > mx ? cj : mx;
  }
  int any_low = 0;
  if (st.flags & 4) {
    mx = block_max_f64(mx, L->shd);
    const double thr = ratio * mx;
    int low = 0;
    for (int j = tid; j < st.K; j += 256) low += ld_l2(st.counts + j) < thr;
    if (low) __hip_atomic_fetch_add((GAS int32_t *)prev + 1, low, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    any_low = __syncthreads_or(low);
  }
  const int any_z = __syncthreads_or((st.flags & 2) ? z : 0);
  if (tid == 0) {
    if (any_low) {
      *halt = st.idx;
      *gp(sdesc.reason) = HAMSM_STOP_LOW;
    } else if (any_z) {
      st.report[0] = 1;
      *halt = st.idx;
      *gp(sdesc.reason) = HAMSM_STOP_ZERO;
    }
  }
}

// The batch rows are static data: the rows a workgroup will need for the stratum's NEXT step are requested while the
// current step's M-step runs and parked in an LDS slab (PF_TILES tiles of 16 rows, row stride 4 KS + 1 elements), so that
// the E-step starts from LDS instead of two dependent global round trips (row index -> row).  Workgroup `erank` of `We`
// owns the tiles erank, erank + We, ...; tiles beyond the slab (fewer workgroups on this XCD than expected) are loaded
// directly.  Slots past D hold the row's last feature and rows past the batch its last row, as k_step_assign reads them.
#define PF_TILES 4
#ifdef PERSIST_FINE
__device__ unsigned long long g_persist_fine[16];
extern "C" int hamsm_debug_persist_fine(unsigned long long *h_out) {
  if (hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_persist_fine), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  unsigned long long z[16] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_persist_fine), z, sizeof z);
  return 0;
}
#define FSTAMP_INIT unsigned long long f_last = wall_clock64();
#define FSTAMP(slot, cond)                                                \
  {                                                                       \
    __builtin_amdgcn_s_waitcnt(0);                                        \
    if ((cond) && threadIdx.x == 0) {                                     \
      const unsigned long long n_ = wall_clock64();                       \
      atomicAdd(&g_persist_fine[slot], n_ - f_last);                      \
      f_last = n_;                                                        \
    }                                                                     \
  }
#else
#define FSTAMP_INIT
#define FSTAMP(slot, cond)
#endif
template <typename XT, int KS>
__device__ __forceinline__ void persist_pf_issue(const StepDev *sp, int erank, int We, int64_t ldx, int D, XT (&pf)[KS]) {
  // thread t: slab row t / 4 (tile t / 64, row (t / 4) % 16), elements (t % 4) + 4 m
  const int tid = threadIdx.x, srow = tid >> 2;
  const long long n_b = sp->n_b;
  const long long tile = (long long)erank + (long long)We * (srow >> 4);
  long long r = tile * 16 + (srow & 15);
  r = r < n_b ? r : n_b - 1;
  r = r < 0 ? 0 : r;
  const GAS int32_t *rows = gp(sp->rows);
  const long long f = rows ? (long long)rows[r] : r;
  const GAS XT *X = (const GAS XT *)gp((const XT *)sp->X);
#pragma unroll
  for (int m = 0; m < KS; m++) {
    const int k = (tid & 3) + 4 * m;
    pf[m] = X[f * ldx + (k < D ? k : D - 1)];
  }
}
template <typename XT, int KS>
__device__ __forceinline__ void persist_pf_store(XT *xa, const XT (&pf)[KS]) {
  const int tid = threadIdx.x, srow = tid >> 2;
#pragma unroll
  for (int m = 0; m < KS; m++) xa[srow * (4 * KS + 1) + (tid & 3) + 4 * m] = pf[m];
}

// E-step: the row tiles erank, erank + We, ... of the step (16 rows each), RT at a time -- the arithmetic of k_step_assign
template <typename XT, int KS, int JB, int RT>
__device__ __attribute__((noinline)) void persist_estep(const StepDev *sp, int erank, int We, int64_t ldx, int D, PersistLds *L,
                                                        const XT *xa) {
  constexpr int RW = 16 * RT;
  static_assert(RW <= 32, "PersistLds::sd holds 32 rows");
  static_assert(PF_TILES % RT == 0, "a pass never straddles the slab's end");
  const StepDev &sdesc = *sp;
  const StepG st(sdesc);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double inf = __builtin_huge_val();
  const GAS XT *X = (const GAS XT *)st.X;
  const int njt = (st.K + 15) >> 4;
  const long long ntile = (st.n_b + 15) >> 4;
  double b[JB][KS], cs[JB][4];
  auto load_group = [&](int t0) {
#pragma unroll
    for (int u = 0; u < JB; u++) {
      const int jt = t0 + 4 * u < njt ? t0 + 4 * u : njt - 1;
#pragma unroll
      for (int q = 0; q < 4; q++) cs[u][q] = ld_l2(st.csq + jt * 16 + (lane >> 4) + 4 * q);
#pragma unroll
      for (int ks = 0; ks < KS; ks++) b[u][ks] = ld_l2(st.Bpack + ((size_t)jt * KS + ks) * 64 + lane);
    }
  };
  const bool one_group = njt <= 4 * JB;  // the stratum's operands stay in registers for all of the workgroup's tiles
  FSTAMP_INIT
  if (one_group) load_group(wv);
  FSTAMP(0, erank == 0)
  // pass p covers this workgroup's tiles number p RT .. p RT + RT - 1 (tile number q = tile erank + q We)
  for (int q0 = 0; (long long)erank + (long long)We * q0 < ntile; q0 += RT) {
    long long tile[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) tile[rt] = (long long)erank + (long long)We * (q0 + rt);
    double a[RT][KS];
    if (q0 < PF_TILES) {
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) a[rt][ks] = (double)xa[((q0 + rt) * 16 + (lane & 15)) * (4 * KS + 1) + ks * 4 + (lane >> 4)];
    } else {
      long long f[RT];
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
        const long long r = tile[rt] * 16 + (lane & 15);
        const long long rc = r < st.n_b ? r : st.n_b - 1;
        f[rt] = st.rows ? (long long)st.rows[rc] : rc;
      }
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
          const int k = ks * 4 + (lane >> 4);
          a[rt][ks] = (double)X[f[rt] * ldx + (k < D ? k : D - 1)];
        }
    }
    double bd[RT];
    int bi[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
      bd[rt] = inf;
      bi[rt] = 0x7fffffff;
    }
    FSTAMP(1, erank == 0)
    for (int t0 = wv; t0 < njt; t0 += 4 * JB) {
      if (!one_group) load_group(t0);
#pragma unroll
      for (int u = 0; u < JB; u++) {
        const int jt = t0 + 4 * u;
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
              bool lt = d < bd[rt];
              bd[rt] = lt ? d : bd[rt];
              bi[rt] = lt ? j0 + 4 * q : bi[rt];
            }
        }
      }
    }
    FSTAMP(2, erank == 0)
    // the batch rows as f64 for the M-step (rows of 4 KS doubles; the buffer is padded to 64-row multiples)
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
      if (((rt + 1) & 3) == wv && tile[rt] < ((st.n_b + 63) >> 6) * 4) {
        const long long r = tile[rt] * 16 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) st.Xg[(size_t)r * (4 * KS) + ks * 4 + (lane >> 4)] = a[rt][ks];
      }
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
      double d = bd[rt];
      int j = bi[rt];
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        double d2 = __shfl_xor(d, off, 64);
        int j2 = __shfl_xor(j, off, 64);
        bool take = (d2 < d) || (d2 == d && j2 < j);
        d = take ? d2 : d;
        j = take ? j2 : j;
      }
      if (lane < 16) {
        L->sd[wv][rt * 16 + lane] = d;
        L->sj[wv][rt * 16 + lane] = j;
      }
    }
    __syncthreads();
    if (tid < RW) {
      const long long r = ((long long)erank + (long long)We * (q0 + (tid >> 4))) * 16 + (tid & 15);
      if (r < st.n_b) {
        double d = L->sd[0][tid];
        int j = L->sj[0][tid];
#pragma unroll
        for (int w = 1; w < 4; w++) {
          double d2 = L->sd[w][tid];
          int j2 = L->sj[w][tid];
          bool take = (d2 < d) || (d2 == d && j2 < j);
          d = take ? d2 : d;
          j = take ? j2 : j;
        }
        if (j == 0x7fffffff) j = 0;
        st.labels[r] = j;
        __hip_atomic_fetch_or(&st.mbits[(size_t)j * sdesc.wstride + (r >> 6)], 1ull << (r & 63), __ATOMIC_RELAXED,
                              PERSIST_ATOMIC_SCOPE);
      }
    }
    __syncthreads();  // sd / sj are reused by the next pass
    FSTAMP(3, erank == 0)
  }
}

// M-step: one wave per centroid (rank * 4 + wave, + 4 W, ...) -- the arithmetic of k_step_update<H = 1>
template <typename XT, int KS>
__device__ __attribute__((noinline)) void persist_mstep(const StepDev *sp, int rank, int W, int D, int *lds_i, const StepDev *sp_next,
                                                        int erank, int We, int64_t ldx, XT *xa) {
  const StepDev &sdesc = *sp;
  const StepG st(sdesc);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // the next step's rows of this workgroup: requested now, parked in the slab when this M-step is done (the E-step that
  // read the slab finished before the barrier in front of this call)
  XT pf[KS];
  FSTAMP_INIT
  if (sp_next) persist_pf_issue<XT, KS>(sp_next, erank, We, ldx, D, pf);
  FSTAMP(4, rank == 0)
  int *list = lds_i + wv * UPD_LIST;
  unsigned *srow = (unsigned *)list;
  double *swl = (double *)(list + UPD_SC);
  const GAS double *Xg = st.Xg;
  const int ldg = 4 * KS;
  const int nw = (int)((st.n_b + 63) >> 6);
  const bool weighted = st.wf != nullptr;
  const int kq = lane < D ? lane : D - 1;
  for (int j = rank * 4 + wv; j < st.K; j += 4 * W) {
    GAS unsigned long long *mb = st.mbits + (size_t)j * sdesc.wstride;
    const double cnt0 = ld_l2(st.counts + j);
    double acc = ld_l2(st.C + (size_t)j * D + kq) * cnt0;  // (KS <= 16: D <= 64, one feature per lane)
    double wsum = 0.0;
    long long total = 0;
    for (int wb = 0; wb < nw; wb += 64) {
      // read and clear in one: the word is ready for the stratum's next step (wstride is a multiple of 64)
      const unsigned long long w = __hip_atomic_exchange(mb + wb + lane, 0ull, __ATOMIC_RELAXED, PERSIST_ATOMIC_SCOPE);
      FSTAMP(5, rank == 0)
      const int pc = __popcll(w);
      int inc = pc;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off, 64);
        inc += lane >= off ? t : 0;
      }
      const int excl = inc - pc, blk = __shfl(inc, 63, 64);
      for (int p0 = 0; p0 < blk; p0 += UPD_SC) {
        {
          unsigned long long t = w;
          int idx = excl - p0;
          while (t) {
            const int bit = __builtin_ctzll(t);
            t &= t - 1;
            if (idx >= 0 && idx < UPD_SC) srow[idx] = (unsigned)((wb + lane) * 64 + bit);
            idx++;
          }
        }
        const int np = blk - p0 < UPD_SC ? blk - p0 : UPD_SC;
        for (int m0 = 0; m0 < np; m0 += 64) {
          const int cm = np - m0 < 64 ? np - m0 : 64;
          if (weighted) {
            if (lane < cm) {
              const long long r = srow[m0 + lane];
              swl[lane] = st.wf[st.rows ? (long long)st.rows[r] : r];
            }
            for (int q = 0; q < cm; q++) wsum = wsum + swl[q];
          }
          // member rows: up to 32 requested before the first add, added in sample order
          for (int q0 = 0; q0 < cm; q0 += 32) {
            double xv[32];
#pragma unroll
            for (int q = 0; q < 32; q++) {
              const int mq = q0 + q < cm ? q0 + q : cm - 1;
              xv[q] = ld_l2(Xg + (size_t)srow[m0 + mq] * ldg + kq);
            }
#pragma unroll
            for (int q = 0; q < 32; q++) {
              const int mq = q0 + q < cm ? q0 + q : cm - 1;
              const double wq = weighted ? swl[mq] : 1.0;
              const double nv = acc + xv[q] * wq;
              acc = (q0 + q < cm) ? nv : acc;
            }
          }
        }
      }
      total += blk;
    }
    FSTAMP(6, rank == 0)
    if (total == 0) continue;
    if (!weighted) wsum = (double)total;
    if (!(wsum > 0)) continue;
    const double cnt1 = cnt0 + wsum;
    const double alpha = 1.0 / cnt1;
    const double v = acc * alpha;
    if (lane < D) {
      st.C[(size_t)j * D + lane] = v;
      st.Bpack[((size_t)(j >> 4) * KS + (lane >> 2)) * 64 + (lane & 3) * 16 + (j & 15)] = v;
    }
    if (lane == 0) st.counts[j] = cnt1;
    double *vb = (double *)list;
    vb[lane] = lane < D ? v : 0.0;
    double a2 = 0.0;
    for (int k = 0; k + 1 < D; k += 2) {
      const double2 t = *(const double2 *)&vb[k];
      a2 = __builtin_fma(t.x, t.x, a2);
      a2 = __builtin_fma(t.y, t.y, a2);
    }
    if (D & 1) a2 = __builtin_fma(vb[D - 1], vb[D - 1], a2);
    if (lane == 0) st.csq[j] = a2;
    FSTAMP(7, rank == 0)
  }
  if (sp_next) persist_pf_store<XT, KS>(xa, pf);
  FSTAMP(8, rank == 0)
}

// re-seeding of low-count centroids from the permutation head the step carries -- k_step_reassign (one workgroup)
template <typename XT, int KS>
__device__ __attribute__((noinline)) void persist_reassign(const StepDev *sp, int64_t ldx, int D, double ratio, PersistLds *L) {
  const StepDev &sdesc = *sp;
  const StepG st(sdesc);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double inf = __builtin_huge_val();
  const GAS XT *X = (const GAS XT *)st.X;
  double mx = -inf;
  for (int j = tid; j < st.K; j += 256) {
    const double cj = ld_l2(st.counts + j);
    mx = cj > mx ? cj : mx;
  }
  mx = block_max_f64(mx, L->shd);
  const double thr = ratio * mx;
  int low = 0;
  double mn = -inf;
  for (int j = tid; j < st.K; j += 256) {
    const double cj = ld_l2(st.counts + j);
    if (cj < thr) low++;
    else mn = -cj > mn ? -cj : mn;
  }
  low = block_sum_i32(low, L->shi);
  const double new_count = -block_max_f64(mn, L->shd);
  if (tid == 0) st.report[1] = low;
  int why = 0;
  if (low == 0) why = HAMSM_STOP_UNUSED;
  else if (low > sdesc.n_perm || (double)low > 0.5 * (double)st.n_b) why = HAMSM_STOP_CAP;
  if (why) {
    if (tid == 0) {
      *gp(sdesc.halt) = st.idx + 1;
      *gp(sdesc.reason) = why;
    }
    return;
  }
  const GAS int32_t *perm = gp(sdesc.perm);
  int base = 0;
  for (int j0 = 0; j0 < st.K; j0 += 256) {
    const int j = j0 + tid;
    const bool flag = j < st.K && ld_l2(st.counts + (j < st.K ? j : 0)) < thr;
    const unsigned long long bal = __ballot(flag);
    const int in_wave = __popcll(bal & ((1ull << lane) - 1));
    __syncthreads();
    if (lane == 0) L->shi[wv] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      before += w < wv ? L->shi[w] : 0;
      total += L->shi[w];
    }
    if (flag) {
      const int e = before + in_wave;
      L->lj[e] = j;
      L->lr[e] = perm[base + e];
    }
    __syncthreads();
    for (int e = wv; e < total; e += 4) {
      const int jc = L->lj[e];
      const long long r = L->lr[e];
      const long long fr = st.rows ? (long long)st.rows[r] : r;
      const double v = (lane < D) ? (double)X[fr * ldx + lane] : 0.0;
      if (lane < D) {
        st.C[(size_t)jc * D + lane] = v;
        st.Bpack[((size_t)(jc >> 4) * KS + (lane >> 2)) * 64 + (lane & 3) * 16 + (jc & 15)] = v;
      }
      double a2 = 0.0;
      for (int k = 0; k < D; k++) {
        const double vk = readlane_f64(v, k);
        a2 = __builtin_fma(vk, vk, a2);
      }
      if (lane == 0) {
        st.csq[jc] = a2;
        st.counts[jc] = new_count;
      }
    }
    base += total;
    __syncthreads();
  }
}

// low-count test of a chain's LAST step when no permutation came with it -- k_step_check
__device__ __attribute__((noinline)) void persist_tailcheck(const StepDev *sp, double ratio, int n_steps, PersistLds *L) {
  const StepDev &sdesc = *sp;
  const StepG st(sdesc);
  const int tid = threadIdx.x;
  double mx = -__builtin_huge_val();
  for (int j = tid; j < st.K; j += 256) {
    const double cj = ld_l2(st.counts + j);
    mx = cj > mx ? cj : mx;
  }
  mx = block_max_f64(mx, L->shd);
  const double thr = ratio * mx;
  int low = 0;
  for (int j = tid; j < st.K; j += 256) low += ld_l2(st.counts + j) < thr;
  low = block_sum_i32(low, L->shi);
  if (tid == 0 && low) {
    st.report[1] = low;
    *gp(sdesc.halt) = n_steps;
    *gp(sdesc.reason) = HAMSM_STOP_LOW;
  }
}

// phase timers of the persistent pass (HAMSM_PERSIST_PROF=1): 100 MHz ticks summed over the steps, as seen by thread 0 of
// every workgroup with rank 0 (one per stratum): [0] tests, [1] E-step, [2] wait at barrier A, [3] M-step, [4] re-seeding,
// [5] wait at barrier B, [6] steps, [7] whole chain
__device__ unsigned long long g_persist_prof[8];
extern "C" int hamsm_debug_persist_profile(unsigned long long *h_out, int reset) {
  if (hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_persist_prof), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_persist_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#define PPROF(slot)                                                              \
  if (prof && rank == 0 && tid == 0) {                                           \
    const unsigned long long now_ = wall_clock64();                              \
    atomicAdd(&g_persist_prof[slot], now_ - t_last);                             \
    t_last = now_;                                                               \
  }

template <typename XT, int KS, int JB>
__global__ __launch_bounds__(256, 2) void k_train_persist(const StepDev *__restrict__ steps, const int *__restrict__ chain_off,
                                                          const int *__restrict__ chain_idx, PersistCtl *ctl_, int S,
                                                          int64_t ldx, int D, double ratio, int n_steps, int prof) {
  constexpr int RT = 2;  // row tiles in flight per workgroup pass
  const int tid = threadIdx.x;
  GAS PersistCtl *ctl = (GAS PersistCtl *)ctl_;
  GAS unsigned *err = &ctl->err;
  __shared__ PersistLds L;
  extern __shared__ int lds_i[];  // M-step strips: 4 x UPD_LIST ints

  // ---- registration ---------------------------------------------------------------------------------
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7;
  if (tid == 0) {
    // (agent scope: every XCD must see these)
    L.sh_i[0] = (int)__hip_atomic_fetch_add(&ctl->nwg_xcd[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&ctl->arrive_all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    const unsigned long long t0 = wall_clock64();
    while (ld_l2(&ctl->arrive_all) < gridDim.x) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > PERSIST_TMO_REG || ld_l2(err) != 0u) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    L.sh_i[1] = ok;
    int n_act = 0, mine = -1;
    for (int x = 0; x < 8; x++) {
      const unsigned nx = ld_l2(&ctl->nwg_xcd[x]);
      if (x == (int)xcc) {
        mine = n_act;
        L.sh_i[2] = (int)nx;
      }
      n_act += nx > 0;
    }
    L.sh_i[3] = n_act;
    L.sh_i[4] = mine;
  }
  __syncthreads();
  if (!L.sh_i[1]) return;
  const int rank = L.sh_i[0], W = L.sh_i[2], n_act = L.sh_i[3], a_idx = L.sh_i[4];
  __syncthreads();

  // workgroup W - 1 of an XCD runs the tests that ride on a step (they need the counts the previous M-step left: 2 us of
  // dependent reductions) instead of row tiles; the tiles are dealt over the other W - 1
  const int We = W > 1 ? W - 1 : 1, erank = rank < We ? rank : -1;
  const bool tester = rank == W - 1;
  XT *xa = (XT *)(lds_i + 4 * UPD_LIST);  // [PF_TILES * 16][4 KS + 1]
  for (int s = a_idx; s < S; s += n_act) {
    GAS unsigned *bar = &ctl->bar[s];
    unsigned epoch = 0;
    const int c0 = chain_off[s], c1 = chain_off[s + 1];
    int last_ci = -1;
    bool stopped = false;
    unsigned long long t_last = wall_clock64();
    const unsigned long long t_chain = t_last;
    // first step of the chain with rows: its slab is filled here
    int ci = c0;
    while (ci < c1 && steps[__builtin_amdgcn_readfirstlane(chain_idx[ci])].n_b <= 0) ci++;
    if (ci < c1 && erank >= 0) {
      XT pf[KS];
      persist_pf_issue<XT, KS>(steps + __builtin_amdgcn_readfirstlane(chain_idx[ci]), erank, We, ldx, D, pf);
      persist_pf_store<XT, KS>(xa, pf);
    }
    __syncthreads();
    for (; ci < c1;) {
      const StepDev *sp = steps + __builtin_amdgcn_readfirstlane(chain_idx[ci]);  // static, uniform: scalar loads
      int cn = ci + 1;  // the stratum's next step with rows (the multi-kernel path launches nothing for empty ones either)
      while (cn < c1 && steps[__builtin_amdgcn_readfirstlane(chain_idx[cn])].n_b <= 0) cn++;
      const StepDev *sp_next = cn < c1 ? steps + __builtin_amdgcn_readfirstlane(chain_idx[cn]) : nullptr;
      last_ci = ci;
      ci = cn;
      GAS int32_t *halt = gp(sp->halt);
      const int idx = sp->idx, flags = sp->flags;
      // the tests that ride on the step (the counts are final since the previous step's barrier) beside the E-step
      if (tester && (flags & 6) && idx < ld_l2(halt)) persist_precheck(sp, ratio, &L);
      PPROF(0)
      if (erank >= 0) persist_estep<XT, KS, JB, RT>(sp, erank, We, ldx, D, &L, xa);
      PPROF(1)
      if (!xcd_barrier(bar, (unsigned)W * ++epoch, err, &L.sh_i[8])) return;
      PPROF(2)
      // the stratum's chain stops here when a test said so: every workgroup reads the same word in the same window
      // (nobody writes it between this barrier and the next)
      if (idx >= ld_l2(halt)) {
        stopped = true;
        break;
      }
      persist_mstep<XT, KS>(sp, rank, W, D, lds_i, erank >= 0 ? sp_next : nullptr, erank, We, ldx, xa);
      PPROF(3)
      if ((flags & 1) && sp->perm) {
        if (!xcd_barrier(bar, (unsigned)W * ++epoch, err, &L.sh_i[8])) return;
        if (rank == 0 && idx < ld_l2(halt)) persist_reassign<XT, KS>(sp, ldx, D, ratio, &L);
        PPROF(4)
      }
      if (!xcd_barrier(bar, (unsigned)W * ++epoch, err, &L.sh_i[8])) return;
      PPROF(5)
      if (prof && rank == 0 && tid == 0) atomicAdd(&g_persist_prof[6], 1ull);
    }
    if (prof && rank == 0 && tid == 0) atomicAdd(&g_persist_prof[7], wall_clock64() - t_chain);
    if (!stopped && last_ci >= 0 && rank == 0) {
      const StepDev *sp = steps + __builtin_amdgcn_readfirstlane(chain_idx[last_ci]);
      if ((sp->flags & 1) && !sp->perm && sp->idx < ld_l2(gp(sp->halt))) persist_tailcheck(sp, ratio, n_steps, &L);
    }
    __syncthreads();
  }
}

template <typename XT, int KS, int RT, bool BYVAL>
static void launch_step_kernels(hamsm_ctx *c, hipStream_t q, const StepDev *d_steps, const StepPack &pk, int ny,
                                long long max_nb, int maxK, int64_t ldx, double ratio, bool any_perm) {
  constexpr int JB = KS <= 4 ? 8 : (KS <= 8 ? 4 : (KS <= 16 ? 2 : 1));
  dim3 ga((unsigned)((max_nb + 16 * RT - 1) / (16 * RT)), (unsigned)ny);
  // every 128th launch group of the context's own queue is sampled: events around the E-step and the M-step launch, and the
  // E-step's own first-wave-in / last-wave-out stamps (slots prepared in bulk by hamsm_step_kernel_ms(reset))
  const bool sample = q == c->stream && !c->capturing && c->ks_stamps && (c->ks_tick++ & 127) == 0 && c->ks_pending.size() < HAMSM_KS_SLOTS;
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
  if (c->D <= 64) {
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

// the M-step (and the device re-seeding) of steps whose E-step ran through the assignment kernels; descriptors by value
template <typename XT>
static void launch_update_only(hamsm_ctx *c, hipStream_t q, const StepPack &pk, int ny, int maxK, int64_t ldx, double ratio, bool any_perm) {
  const size_t lds = sizeof(int) * (4 * UPD_LIST);
  const dim3 gu((unsigned)((maxK + 3) / 4), (unsigned)ny);
  if (c->D <= 64) hipLaunchKernelGGL((k_step_update<XT, 1, true, 4>), gu, dim3(256), lds, q, (const StepDev *)nullptr, pk, ldx, c->D, c->KS);
  else hipLaunchKernelGGL((k_step_update<XT, 2, true, 4>), gu, dim3(256), lds, q, (const StepDev *)nullptr, pk, ldx, c->D, c->KS);
  if (any_perm)
    hipLaunchKernelGGL((k_step_reassign<XT, true>), dim3((unsigned)ny), dim3(256), 0, q, (const StepDev *)nullptr, pk, ldx, c->D, c->KS, ratio);
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

template <typename XT>
static int launch_persist(hamsm_ctx *c, const StepDev *d_steps, const int *d_coff, const int *d_cidx, PersistCtl *d_ctl,
                          int64_t ldx, double ratio, int n_steps) {
  const size_t lds = sizeof(int) * (4 * UPD_LIST) + sizeof(XT) * (PF_TILES * 16) * (4 * (size_t)c->KS + 1);
  const dim3 grid((unsigned)(8 * c->persist_wpx)), block(256);
  static const int prof = getenv("HAMSM_PERSIST_PROF") ? atoi(getenv("HAMSM_PERSIST_PROF")) : 0;
  switch (c->KS) {
#define CASE(KSV, JBV)                                                                                                   \
  case KSV:                                                                                                              \
    hipLaunchKernelGGL((k_train_persist<XT, KSV, JBV>), grid, block, lds, c->stream, d_steps, d_coff, d_cidx, d_ctl, c->S, ldx, \
                       c->D, ratio, n_steps, prof);                                                                      \
    break;
    CASE(1, 8) CASE(2, 8) CASE(3, 8) CASE(4, 8) CASE(6, 4) CASE(8, 4) CASE(12, 2) CASE(16, 2)
#undef CASE
    default: return hamsm_fail(c, HAMSM_ENOTSUP, "persistent training: KS=%d", c->KS);
  }
  HIPCHK(c, hipGetLastError());
  return HAMSM_OK;
}

// the E-steps of one look-ahead block (descriptors in memory: a throughput kernel, one dependent load does not matter)
template <typename XT>
static int launch_assign_ahead(hamsm_ctx *c, hipStream_t q, const StepDev *d_steps, const AheadDev *d_aheads, const int *d_ylist,
                               int ny, long long max_nb, int64_t ldx) {
  if (ny <= 0 || max_nb <= 0) return HAMSM_OK;
  const dim3 grid((unsigned)((max_nb + 15) / 16), (unsigned)ny);
  switch (c->KS) {
#define CASE(KSV, JBV)                                                                                                  \
  case KSV:                                                                                                             \
    hipLaunchKernelGGL((k_step_assign_ahead<XT, KSV, JBV, 1>), grid, dim3(256), 0, q, d_steps, d_aheads, d_ylist, ldx, c->D); \
    break;
    CASE(1, 8) CASE(2, 8) CASE(3, 8) CASE(4, 8) CASE(6, 4) CASE(8, 4) CASE(12, 2) CASE(16, 2) CASE(24, 1) CASE(32, 1)
#undef CASE
    default: return hamsm_fail(c, HAMSM_ENOTSUP, "look-ahead E-step: KS=%d", c->KS);
  }
  return HAMSM_OK;
}

// end of a pipelined segment whose last M-step wrote the alternate tables: they go home -- unless the stratum's chain stopped
// inside the segment (then the alternate tables may be the OLDER ones; the host sorts that out once it knows where it stopped)
__global__ void k_tables_home(const int32_t *__restrict__ halt, int last_idx, const double *__restrict__ sC, double *__restrict__ dC,
                              size_t nC, const double *__restrict__ sq, double *__restrict__ dq, const double *__restrict__ sn,
                              double *__restrict__ dn, size_t nk, const double *__restrict__ sB, double *__restrict__ dB, size_t nB) {
  if (*halt <= last_idx) return;
  const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = t; i < nC; i += stride) dC[i] = sC[i];
  for (size_t i = t; i < nk; i += stride) {
    dq[i] = sq[i];
    dn[i] = sn[i];
  }
  for (size_t i = t; i < nB; i += stride) dB[i] = sB[i];
}

// one launch of the pipeline: n_m M-steps (by value) and n_e E-steps (descriptors in memory)
template <typename XT>
static int launch_pipe(hamsm_ctx *c, hipStream_t q, const StepPack &pk, const AheadPack &apk, int n_m, int maxK, const StepDev *d_sde,
                       const AheadDev *d_aheads, const int *d_ylist, int n_e, long long e_nb, int64_t ldx, double ratio) {
  const int rt = c->ahead_rt == 2 && c->KS <= 16 ? 2 : 1;
  const unsigned gx_m = (unsigned)((maxK + 3) / 4), gx_e = (unsigned)((e_nb + 16 * rt - 1) / (16 * rt));
  const dim3 grid(gx_m > gx_e ? gx_m : gx_e, (unsigned)(n_m + n_e));
  const size_t lds = sizeof(int) * (4 * AHEAD_STRIP) + sizeof(double) * (4 * AHEAD_HX) + sizeof(int) * (4 * 2 * AHEAD_ITEMS);
  switch (c->KS) {
#define CASE(KSV, JBV, HV)                                                                                                     \
  case KSV:                                                                                                                    \
    if (rt == 2 && KSV <= 16)                                                                                                  \
      hipLaunchKernelGGL((k_step_pipe<XT, KSV, JBV, HV, (KSV <= 16 ? 2 : 1)>), grid, dim3(256), lds, q, pk, apk, n_m, d_sde, d_aheads, \
                         d_ylist, ldx, c->D, ratio);                                                                           \
    else                                                                                                                       \
      hipLaunchKernelGGL((k_step_pipe<XT, KSV, JBV, HV, 1>), grid, dim3(256), lds, q, pk, apk, n_m, d_sde, d_aheads, d_ylist, ldx, \
                         c->D, ratio);                                                                                         \
    break;
    CASE(1, 8, 1) CASE(2, 8, 1) CASE(3, 8, 1) CASE(4, 4, 1) CASE(6, 2, 1) CASE(8, 2, 1) CASE(12, 1, 1) CASE(16, 1, 1) CASE(24, 1, 2) CASE(32, 1, 2)
#undef CASE
    default: return hamsm_fail(c, HAMSM_ENOTSUP, "pipelined training: KS=%d", c->KS);
  }
  return HAMSM_OK;
}

// steps [0,n): consecutive steps of distinct strata share launches (grid.y)
static int run_steps(hamsm_ctx *c, const void *X, int dt, int64_t ldx, const int32_t *d_rows, const double *d_w,
                     const hamsm_step *h_steps, int64_t n_steps, double ratio, const int32_t *h_perm, int64_t n_perm_total,
                     int32_t *h_report, int64_t *h_resume, int32_t *h_reason, int want_zero_flag,
                     int32_t **labels_of_first) {
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
  // Groups whose E-step is heavy enough to be bound by the f64 matrix unit run it through the assignment kernels instead
  // (hamsm_assign_rows: ranking on the bf16 / f32 matrix unit under a proven bound, exact f64 kernel for the rest -- the same
  // labels): the group's strata ascend, their row lists lie back to back, the steps share one frame
  // span and say how long it is (`n_span`).  Such a pass keeps to one queue (the assignment kernels have one scratch arena).
  std::vector<char> ranked(gstart.size() - 1, 0);
  bool any_ranked = false;
  int64_t span_max = 0;
  if (c->train_ranked && !generic && d_rows && h_report && c->assign_filter && c->have_mapper)
    for (size_t g = 0; g + 1 < gstart.size(); g++) {
      const int64_t i = gstart[g], j = gstart[g + 1];
      bool good = h_steps[i].n_span > 0;
      double flop = 0;
      for (int64_t q = i; q < j && good; q++) {
        const hamsm_step &st_ = h_steps[q];
        good = st_.n_b > 0 && st_.x_off == h_steps[i].x_off && st_.n_span == h_steps[i].n_span && !(st_.flags & HAMSM_STEP_FIRST);
        if (q > i) good = good && st_.stratum > h_steps[q - 1].stratum && st_.rows_off == h_steps[q - 1].rows_off + h_steps[q - 1].n_b;
        flop += 2.0 * (double)st_.n_b * c->st[st_.stratum].K * c->D;
      }
      if (good && (c->train_ranked == 2 || flop >= c->ranked_min_flop)) {
        ranked[g] = 1;
        any_ranked = true;
        span_max = h_steps[i].n_span > span_max ? h_steps[i].n_span : span_max;
      }
    }
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return o;
  };
  size_t o_gcnt = take(any_ranked ? sizeof(unsigned long long) * (size_t)c->S * (gstart.size() - 1) : 0),
         o_lspan = take(sizeof(int32_t) * (size_t)span_max);
  size_t o_steps = take(sizeof(StepDev) * n_steps), o_rep = take(sizeof(int32_t) * 2 * n_steps),
         o_lab = take(sizeof(int32_t) * (size_t)lab_hi), o_halt = take(sizeof(int32_t) * 2 * (size_t)c->S),
         o_perm = take(sizeof(int32_t) * (size_t)(n_perm_total > 0 ? n_perm_total : 1));
  // the gathered batch copy and the membership bitmaps are reused by every launch group: one slot per stratum position,
  // stream-ordered
  const int nslot = c->S < 65535 ? c->S : 65535;
  const size_t xg_row = sizeof(double) * 4 * (size_t)c->KS;
  const size_t nb_pad = ((size_t)nb_max + 63) & ~(size_t)63;  // whole E-step tiles (up to 4 x 16 rows)
  // look-ahead blocks (k_step_assign_ahead): a planned pass with row lists, a scratch slot per stratum, batches of at most 4096
  // rows (one block of bitmap words per centroid), at most 128 features.  Every position of a block has its own batch copy,
  // bitmaps and table of undecided rows; steps of the step-by-step path use position 0's.
  const bool ahead_can0 = c->ahead && c->ahead_gmax >= 1 && !generic && d_rows && h_report && nslot == c->S && nb_max > 0 &&
                         nb_max <= 4096 && !c->train_persist;  // (and at most AHEAD_KMAX centroids per stratum: below)
  int Kmax = 0;
  for (int s = 0; s < c->S; s++) Kmax = c->st[s].K > Kmax ? c->st[s].K : Kmax;
  const bool ahead_can = ahead_can0 && Kmax <= AHEAD_KMAX;
  const int npos = ahead_can ? 4 : 1;  // a step's E-step output lives until its M-step, two launches later: four rotating regions
  size_t o_xg = take(generic ? 0 : xg_row * (size_t)nslot * npos * nb_pad);  // (the generic M-step reads the frames in place)
  const int wstride = (int)((((nb_max + 63) >> 6) + 63) & ~(int64_t)63);  // bitmap words per centroid
  const size_t mb_bytes = sizeof(unsigned long long) * (size_t)nslot * npos * Kmax * wstride;
  size_t o_mb = take(mb_bytes);
  const size_t ah_words = ahead_can ? (size_t)n_steps : 0;  // per step: displacement slot, bound, two counters
  const int ah_kst = (Kmax + 63) & ~63;
  const int ah_dp = (c->D + 1) & ~1;
  int ah_hcap = AHEAD_HX / ah_dp < AHEAD_HCAP ? AHEAD_HX / ah_dp : AHEAD_HCAP;  // (>= 5: D <= 128)
  if (const char *v = getenv("HAMSM_AHEAD_HCAP")) ah_hcap = atoi(v) < ah_hcap ? (atoi(v) < 0 ? 0 : atoi(v)) : ah_hcap;  // (experiments)
  size_t o_slot = take(sizeof(double) * (AHEAD_SLOTS + 1) * ah_words), o_cnt = take(sizeof(int32_t) * 4 * ah_words),
         o_hl = take(ahead_can ? sizeof(int32_t) * AHEAD_HCAP * (size_t)nslot * npos : 0),
         o_hd = take(ahead_can ? sizeof(double) * AHEAD_HCAP * (size_t)ah_kst * nslot * npos : 0),
         o_td = take(ahead_can ? sizeof(double) * 3 * AHEAD_TAB_CAP * (size_t)nslot * npos : 0),
         o_tab = take(ahead_can ? sizeof(SuspEntry) * AHEAD_TAB_CAP * (size_t)nslot * npos : 0),
         o_ylist = take(sizeof(int) * ah_words),
         o_ah = take(sizeof(AheadDev) * ah_words), o_sde = take(sizeof(StepDev) * ah_words), o_tail = take(sizeof(StepDev) * (size_t)c->S),
         o_alt = take(ahead_can ? sizeof(double) * (size_t)c->S * ((size_t)((Kmax + 15) & ~15) * (c->D + 2 + c->KS * 4)) : 0);
  // persistent pass: control block (registration, error word, one barrier counter per stratum) and the strata's chains
  const size_t ctl_bytes = (sizeof(PersistCtl) + sizeof(unsigned) * (size_t)c->S + 15) & ~(size_t)15;
  size_t o_ctl = take(ctl_bytes), o_coff = take(sizeof(int) * ((size_t)c->S + 1)), o_cidx = take(sizeof(int) * (size_t)n_steps);
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
  if (n_steps >= 0x7f000000) return hamsm_fail(c, HAMSM_EINVAL, "too many steps");
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
      sd[j0].Xg = (double *)(b + o_xg + slot * npos * nb_pad * xg_row);
      sd[j0].mbits = (unsigned long long *)(b + o_mb) + slot * npos * Kmax * wstride;
      sd[j0].wstride = wstride;
    }
  if (any_ranked) {
    // the strata's row counts of every group (what the assignment kernels' bucket passes would have counted)
    rc = hamsm_refresh_table(c);
    if (rc) return rc;
    c->h_stage.assign(sizeof(unsigned long long) * (size_t)c->S * (gstart.size() - 1), 0);
    unsigned long long *gc = (unsigned long long *)c->h_stage.data();
    for (size_t g = 0; g + 1 < gstart.size(); g++)
      if (ranked[g])
        for (int64_t q = gstart[g]; q < gstart[g + 1]; q++) gc[g * c->S + h_steps[q].stratum] = (unsigned long long)h_steps[q].n_b;
    HIPCHK(c, hipMemcpyAsync(b + o_gcnt, gc, c->h_stage.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (hamsm_assign_rows re-uses the staging vector)
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
  // One persistent launch for the whole run (§5.3a) when the run is a planned pass (row lists, a scratch slot per stratum,
  // more than one launch group) of at most 64 features, when HAMSM_TRAIN_PERSIST=1 asks for it (off by default: §5.3a).
  bool persist_done = false;
  if (c->train_persist && h_report && d_rows && nslot == c->S && c->KS <= 16 && gstart.size() > 2 && n_steps < 0x7fffffff) {
    std::vector<int> coff(c->S + 1, 0), cidx((size_t)n_steps);
    for (int64_t i = 0; i < n_steps; i++) coff[h_steps[i].stratum + 1]++;
    for (int s = 0; s < c->S; s++) coff[s + 1] += coff[s];
    {
      std::vector<int> fill(coff.begin(), coff.end() - 1);
      for (int64_t i = 0; i < n_steps; i++) cidx[fill[h_steps[i].stratum]++] = (int)i;
    }
    HIPCHK(c, hipMemcpyAsync(b + o_coff, coff.data(), sizeof(int) * coff.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b + o_cidx, cidx.data(), sizeof(int) * cidx.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(b + o_ctl, 0, ctl_bytes, c->stream));
    hamsm_timer_begin(c);
    rc = (dt == HAMSM_F32) ? launch_persist<float>(c, d_steps, (const int *)(b + o_coff), (const int *)(b + o_cidx),
                                                   (PersistCtl *)(b + o_ctl), ldx, ratio, (int)n_steps)
                           : launch_persist<double>(c, d_steps, (const int *)(b + o_coff), (const int *)(b + o_cidx),
                                                    (PersistCtl *)(b + o_ctl), ldx, ratio, (int)n_steps);
    hamsm_timer_end(c, ST_STEP);
    if (rc) {
      hipDeviceSynchronize();
      return rc;
    }
    unsigned perr = 0;
    HIPCHK(c, hipMemcpyAsync(&perr, b + o_ctl + offsetof(PersistCtl, err), sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (also: coff / cidx / sd are locals the uploads read)
    if (perr == 0) persist_done = true;
    else if (perr == 1) {
      // not every workgroup of the grid became resident: nothing was modified; this context keeps to the multi-kernel path
      c->train_persist = 0;
      fprintf(stderr, "[hamsm] persistent training launch could not register all workgroups: using the multi-kernel path\n");
    } else
      return hamsm_fail(c, HAMSM_EDEVICE, "persistent training kernel: a barrier timed out (ctl.err=%u)", perr);
  }
  // what each stratum's chain has come to: taken from the stop words after every pass over the launches.  A pass ends for a
  // stratum when all its steps are applied or a prediction of the host's plan failed (HAMSM_STOP_*: reported to the caller),
  // or when a look-ahead block was given up (HAMSM_STOP_AHEAD: never reported; the next pass relaunches from that step).
  std::vector<int32_t> fin_halt(c->S, 0x7f7f7f7f), fin_reason(c->S, 0);
  if (!persist_done) {
  std::vector<int64_t> from(c->S, 0);  // steps of stratum s below from[s] are applied
  std::vector<char> live(c->S, 1);
  bool ahead_on = ahead_can && gstart.size() > 2;
  int ahead_fails = 0;
  const bool dbg_enq = getenv("HAMSM_DEBUG_ENQUEUE") != nullptr;
  const bool dbg_ahead = getenv("HAMSM_DEBUG_AHEAD") != nullptr;
  for (int pass = 0;; pass++) {
  hamsm_timer_begin(c);
  // lanes: stratum s always runs on queue s % n_lanes, so every stratum's chain stays ordered; a lane's part of a
  // launch group travels by value (<= PACK_NY strata) -- otherwise one queue and descriptors in memory
  int n_lanes = c->step_lanes;
  for (size_t g = 0; g + 1 < gstart.size() && n_lanes > 1; g++) {
    std::vector<int> cnt(n_lanes, 0);
    for (int64_t q = gstart[g]; q < gstart[g + 1]; q++)
      if (++cnt[h_steps[q].stratum % n_lanes] > PACK_NY) n_lanes = 1;
  }
  if (gstart.size() <= 2 || c->S < 2 || nslot != c->S || !d_rows || generic || any_ranked) n_lanes = 1;  // (labels scratch is per step only with row lists)
  auto runs = [&](int64_t q) { return live[h_steps[q].stratum] && q >= from[h_steps[q].stratum]; };
  auto lane_of = [&](int64_t q) { return n_lanes > 1 ? h_steps[q].stratum % n_lanes : 0; };
  auto lane_of_stratum = [&](int y) { return n_lanes > 1 ? y % n_lanes : 0; };
  // ---- plan the pipelined segments of this pass (host only; fills the steps' look-ahead fields before they are uploaded) ----
  // Per lane: the first `n0` launch groups step by step (a relaunch after a failed premise: none), then SEGMENTS: maximal runs
  // of launch groups in which the lane's strata are the same, every step has rows, none is a first step after seeding and
  // none carries a re-seeding (its centroids jump).  A segment of groups a .. z runs as
  //     E{a exact}   M{a}   E{a+1 exact, a+2}   M{a+1} + E{a+3}   M{a+2} + E{a+4}   ...   M{z-1}   M{z}
  // -- from the fourth launch on ONE launch per WE iteration: the M-step of iteration i beside the E-step of iteration i + 2,
  // which reads the centroid tables that M-step reads (it writes the other buffer).  Labels of i + 2 are therefore two updates
  // old when they are used; M{i} measures exactly that displacement for M{i+1} to check, and evaluates the rows iteration
  // i + 1 left undecided or open against the centroids it has just written.
  struct PipeOp {     // one launch of a lane
    size_t g;         // issued when the launch loop reaches this group (the M part's group; for E-only launches: the next M's)
    int order;        // position among the launches issued at the same group
    std::vector<int64_t> m;  // steps whose M-step runs (the lane's steps of one group), possibly none
    int y0, y1;       // E part: ylist[y0 .. y1)
    long long e_nb;   // largest batch of the E part
  };
  const size_t NG = gstart.size() - 1;
  std::vector<std::vector<PipeOp>> ops(n_lanes);
  std::vector<int> ylist;
  std::vector<char> in_seg(NG * (size_t)n_lanes, 0);  // [g * n_lanes + l]: the lane's steps of this group run inside a segment
  std::vector<AheadDev> ahd(ahead_can ? (size_t)n_steps : 0);
  std::vector<char> out_alt((size_t)n_steps, 0);      // the step's M-step writes the alternate tables (else the stratum's own)
  struct CopyBack {
    size_t g;
    int y;
    int64_t q;  // the segment's last step of the stratum
  };
  std::vector<CopyBack> copy_back;  // after this group's launches the stratum's alternate tables go home
  const size_t kp = (size_t)((Kmax + 15) & ~15);
  const size_t alt_stride = kp * c->D + 2 * kp + kp * c->KS * 4;  // doubles per stratum: C, csq, counts, Bpack
  auto altC = [&](int y) { return (double *)(b + o_alt) + (size_t)y * alt_stride; };
  auto altCsq = [&](int y) { return altC(y) + kp * c->D; };
  auto altCnt = [&](int y) { return altCsq(y) + kp; };
  auto altBp = [&](int y) { return altCnt(y) + kp; };
  if (ahead_can)
    for (int64_t i = 0; i < n_steps; i++) {  // (a relaunch plans afresh: back to region 0 of the stratum, tables at home)
      memset(&ahd[i], 0, sizeof(AheadDev));
      ahd[i].pos = -1;
      const Stratum &t = c->st[h_steps[i].stratum];
      const size_t slot = (size_t)h_steps[i].stratum;
      sd[i].Xg = (double *)(b + o_xg + slot * npos * nb_pad * xg_row);
      sd[i].mbits = (unsigned long long *)(b + o_mb) + slot * npos * Kmax * wstride;
      sd[i].C = t.C;
      sd[i].counts = t.counts;
      sd[i].csq = t.csq;
      sd[i].Bpack = t.Bpack;
    }
  // the E-steps read their descriptors from memory: one array of StepDev copies whose table pointers name the buffer the
  // E-step reads (the M-step's copy of the same step, sent by value, names the buffer IT reads)
  std::vector<StepDev> sdE(ahead_can ? (size_t)n_steps : 0);
  if (ahead_on) {
    for (int l = 0; l < n_lanes; l++) {
      int seen_groups = 0;
      const int n0 = pass == 0 ? c->ahead_n0 : 0;
      // the lane's steps per group, and whether the group may be part of a segment
      std::vector<std::vector<int64_t>> mine(NG);
      std::vector<char> ok(NG, 0);
      for (size_t g = 0; g < NG; g++) {
        bool good = true;
        for (int64_t q = gstart[g]; q < gstart[g + 1]; q++)
          if (runs(q) && lane_of(q) == l) {
            mine[g].push_back(q);
            good = good && h_steps[q].n_b > 0 && !(h_steps[q].flags & HAMSM_STEP_FIRST) && !sd[q].perm;
          }
        ok[g] = good && !mine[g].empty();
      }
      auto same_strata = [&](size_t g1, size_t g2) {
        if (mine[g1].size() != mine[g2].size()) return false;
        for (size_t k = 0; k < mine[g1].size(); k++)
          if (h_steps[mine[g1][k]].stratum != h_steps[mine[g2][k]].stratum) return false;
        return true;
      };
      size_t g = 0;
      while (g < NG) {
        if (mine[g].empty()) {
          g++;
          continue;
        }
        if (seen_groups < n0 || !ok[g]) {
          seen_groups++;
          g++;
          continue;
        }
        // the segment's groups (those of the lane that are not empty), at most `seg_max`
        std::vector<size_t> G_;
        size_t ge = g;
        while (ge < NG && (int)G_.size() < c->ahead_gmax * 64) {
          if (mine[ge].empty()) {
            ge++;
            continue;
          }
          if (!ok[ge] || !same_strata(g, ge)) break;
          G_.push_back(ge);
          ge++;
        }
        if (G_.size() < 4) {  // too short to pay for the three launches of a start
          seen_groups += (int)G_.size();
          g = G_.empty() ? g + 1 : G_.back() + 1;
          continue;
        }
        const int n = (int)G_.size(), ny = (int)mine[g].size();
        for (int k = 0; k < n; k++) in_seg[G_[k] * n_lanes + l] = 1;
        // per step: regions, buffers, slots
        for (int k = 0; k < n; k++)
          for (int q_ = 0; q_ < ny; q_++) {
            const int64_t q = mine[G_[k]][q_];
            const int y = h_steps[q].stratum;
            const Stratum &t = c->st[y];
            AheadDev &d = ahd[q];
            const size_t reg = (size_t)y * npos + (size_t)(k & 3);
            d.pos = (k <= 1) ? 0 : 1;  // the segment's first two steps take their labels exactly
            sd[q].Xg = (double *)(b + o_xg + reg * nb_pad * xg_row);
            sd[q].mbits = (unsigned long long *)(b + o_mb) + reg * Kmax * wstride;
            d.slot = (double *)(b + o_slot) + (AHEAD_SLOTS + 1) * q;
            d.bound = (double *)(b + o_slot) + (AHEAD_SLOTS + 1) * q + AHEAD_SLOTS;
            d.tab = (SuspEntry *)(b + o_tab) + reg * AHEAD_TAB_CAP;
            d.cnt = (int32_t *)(b + o_cnt) + 4 * q;
            d.hl = (int32_t *)(b + o_hl) + reg * AHEAD_HCAP;
            d.hd = (double *)(b + o_hd) + reg * AHEAD_HCAP * ah_kst;
            d.td = (double *)(b + o_td) + reg * 3 * AHEAD_TAB_CAP;
            d.hcap = ah_hcap;
            d.kst = ah_kst;
            // tables: M{k} reads buffer k & 1 (0 = the stratum's own), writes the other
            const bool in_alt = k & 1;
            sd[q].C = in_alt ? altC(y) : t.C;
            sd[q].counts = in_alt ? altCnt(y) : t.counts;
            sd[q].csq = in_alt ? altCsq(y) : t.csq;
            sd[q].Bpack = in_alt ? altBp(y) : t.Bpack;
            d.C_out = in_alt ? t.C : altC(y);
            d.counts_out = in_alt ? t.counts : altCnt(y);
            d.csq_out = in_alt ? t.csq : altCsq(y);
            d.Bpack_out = in_alt ? t.Bpack : altBp(y);
            out_alt[q] = !in_alt;
            d.disp_in = k <= 1;  // labels of k + 1 are one update old for k = 1 (and exact for k = 0), two from then on
            // labels of step k were taken against the tables M{k-2} read (k >= 3), M{k-1} read (k = 2), or its own (k <= 1)
            sdE[q] = sd[q];
            if (k >= 2) {
              const bool e_alt = (k == 2) ? ((k - 1) & 1) : ((k - 2) & 1);
              sdE[q].C = e_alt ? altC(y) : t.C;
              sdE[q].counts = e_alt ? altCnt(y) : t.counts;
              sdE[q].csq = e_alt ? altCsq(y) : t.csq;
              sdE[q].Bpack = e_alt ? altBp(y) : t.Bpack;
              // the premise M{k} checks: what M{k-1} measured <= what the E-step assumed
              const int64_t qp = mine[G_[k - 1]][q_];
              d.prev_slot = (const double *)(b + o_slot) + (AHEAD_SLOTS + 1) * qp;
              // the E-step's guess: alpha x the latest measurement at its launch, scaled to the updates its labels will be old
              //   k = 2: launched after M{0} (one update measured), labels one update old
              //   k = 3: launched with M{1}: latest = M{0}'s (one update), labels two updates old
              //   k = 4: launched with M{2}: latest = M{1}'s (one update), labels two updates old
              //   k >= 5: launched with M{k-2}: latest = M{k-3}'s (two updates), labels two updates old
              const int kr = (k == 2) ? 0 : (k == 3 ? 0 : k - 3);
              const int64_t qr = mine[G_[kr]][q_];
              d.ref_slot = (const double *)(b + o_slot) + (AHEAD_SLOTS + 1) * qr;
              d.ref_scale = c->ahead_alpha * ((k == 3 || k == 4) ? 2.0 : 1.0);
              // M{k-1} evaluates this step's undecided and open rows
              AheadDev &pv = ahd[qp];
              pv.nx_cnt = d.cnt;
              pv.nx_hl = d.hl;
              pv.nx_Xg = sd[q].Xg;
              pv.nx_hd = d.hd;
              pv.nx_tab = d.tab;
              pv.nx_td = d.td;
            }
          }
        // the launches
        auto e_list = [&](std::initializer_list<int> ks, int &y0, int &y1, long long &nb) {
          y0 = (int)ylist.size();
          nb = 0;
          for (int k : ks)
            if (k < n)
              for (int64_t q : mine[G_[k]]) {
                ylist.push_back((int)q);
                nb = h_steps[q].n_b > nb ? h_steps[q].n_b : nb;
              }
          y1 = (int)ylist.size();
        };
        {
          PipeOp o{G_[0], 0, {}, 0, 0, 0};
          e_list({0}, o.y0, o.y1, o.e_nb);
          ops[l].push_back(o);
        }
        {
          PipeOp o{G_[0], 1, mine[G_[0]], 0, 0, 0};
          ops[l].push_back(o);
        }
        {
          PipeOp o{G_[1], 0, {}, 0, 0, 0};
          e_list({1, 2}, o.y0, o.y1, o.e_nb);
          ops[l].push_back(o);
        }
        for (int k = 1; k < n; k++) {
          PipeOp o{G_[k], 1, mine[G_[k]], 0, 0, 0};
          e_list({k + 2}, o.y0, o.y1, o.e_nb);
          ops[l].push_back(o);
        }
        if ((n - 1) % 2 == 0)  // M{n-1} reads buffer (n-1) & 1 = 0 and writes the alternate: bring the tables home
          for (int64_t q : mine[G_[n - 1]]) copy_back.push_back({G_[n - 1], h_steps[q].stratum, q});
        c->ahead_stat[0]++;
        c->ahead_stat[1] += (long long)n * ny;
        seen_groups += n;
        g = G_.back() + 1;
      }
    }
  }
  if (pass > 0 || ahead_on) {
    HIPCHK(c, hipMemcpyAsync(b + o_steps, sd.data(), sizeof(StepDev) * n_steps, hipMemcpyHostToDevice, c->stream));
    if (ahead_can) {
      HIPCHK(c, hipMemsetAsync(b + o_slot, 0, sizeof(double) * (AHEAD_SLOTS + 1) * ah_words, c->stream));
      HIPCHK(c, hipMemsetAsync(b + o_cnt, 0, sizeof(int32_t) * 4 * ah_words, c->stream));
      if (!ylist.empty()) {
        HIPCHK(c, hipMemcpyAsync(b + o_ylist, ylist.data(), sizeof(int) * ylist.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(b + o_ah, ahd.data(), sizeof(AheadDev) * ahd.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(b + o_sde, sdE.data(), sizeof(StepDev) * sdE.size(), hipMemcpyHostToDevice, c->stream));
        // the alternate tables start as copies of the stratum's own (what matters is the padding: +inf in csq, zeros in the
        // fragments; every live row is written by the first M-step that writes them)
        for (int y = 0; y < c->S; y++) {
          const Stratum &t = c->st[y];
          if (t.K == 0 || !live[y]) continue;
          const size_t kq = (size_t)((t.K + 15) & ~15);
          HIPCHK(c, hipMemcpyAsync(altCsq(y), t.csq, sizeof(double) * kq, hipMemcpyDeviceToDevice, c->stream));
          HIPCHK(c, hipMemcpyAsync(altBp(y), t.Bpack, sizeof(double) * kq * c->KS * 4, hipMemcpyDeviceToDevice, c->stream));
        }
      }
    }
  }
  hipStream_t lanes[4] = {c->stream, nullptr, nullptr, nullptr};
  // HAMSM_TRAIN_GRAPH=1 (experiment): the pass's launches are captured into a hipGraph -- the side queues join
  // the capture through the fork / join events -- instantiated and launched once, instead of being enqueued one by one
  const bool graph = c->train_graph && pass == 0 && !ahead_on && !any_ranked && !generic && gstart.size() > 2;
  timespec tg0, tg1, tg2, tg3;
  if (graph) {
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (uploads above done: the capture holds kernels only)
    clock_gettime(CLOCK_MONOTONIC, &tg0);
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
  }
  c->capturing = graph;
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
  for (size_t g = 0; g + 1 < gstart.size(); g++) {
    const int64_t i = gstart[g], j = gstart[g + 1];
    for (int l = 0; l < n_lanes; l++) {
      long long max_nb = 0;
      int maxK = 0, ny = 0;
      bool any_perm = false;
      std::vector<int64_t> mine;
      for (int64_t q = i; q < j; q++) {
        if (!runs(q) || lane_of(q) != l) continue;
        mine.push_back(q);
        any_perm = any_perm || sd[q].perm;
        if (h_steps[q].n_b > max_nb) max_nb = h_steps[q].n_b;
        if (c->st[h_steps[q].stratum].K > maxK) maxK = c->st[h_steps[q].stratum].K;
      }
      ny = (int)mine.size();
      if (ny == 0 || max_nb <= 0) continue;
      if (ahead_on && in_seg[g * n_lanes + l]) {
        // a group inside a pipelined segment: the launches planned for it, in order
        const double ratio_m = ratio;
        for (const PipeOp &o : ops[l]) {
          if (o.g != g) continue;
          const int n_e = o.y1 - o.y0;
          const int *d_yl = (const int *)(b + o_ylist) + o.y0;
          if (o.m.empty()) {
            rc = (dt == HAMSM_F32) ? launch_assign_ahead<float>(c, lanes[l], (const StepDev *)(b + o_sde), (const AheadDev *)(b + o_ah), d_yl, n_e, o.e_nb, ldx)
                                   : launch_assign_ahead<double>(c, lanes[l], (const StepDev *)(b + o_sde), (const AheadDev *)(b + o_ah), d_yl, n_e, o.e_nb, ldx);
          } else {
            for (size_t c0_ = 0; c0_ < o.m.size() && !rc; c0_ += PACK_NY) {  // (descriptors by value: PACK_NY strata per launch)
              const int n_m = (int)(o.m.size() - c0_ < PACK_NY ? o.m.size() - c0_ : PACK_NY);
              AheadPack apk;
              int mK = 0;
              for (int q = 0; q < n_m; q++) {
                pk.s[q] = sd[o.m[c0_ + q]];
                apk.a[q] = ahd[o.m[c0_ + q]];
                mK = pk.s[q].K > mK ? pk.s[q].K : mK;
              }
              for (int q = n_m; q < PACK_NY; q++) apk.a[q] = apk.a[0];
              pk.n_next = 0;
              const bool with_e = c0_ == 0 && n_e > 0;
              rc = (dt == HAMSM_F32) ? launch_pipe<float>(c, lanes[l], pk, apk, n_m, mK, (const StepDev *)(b + o_sde), (const AheadDev *)(b + o_ah),
                                                          d_yl, with_e ? n_e : 0, with_e ? o.e_nb : 0, ldx, ratio_m)
                                     : launch_pipe<double>(c, lanes[l], pk, apk, n_m, mK, (const StepDev *)(b + o_sde), (const AheadDev *)(b + o_ah),
                                                           d_yl, with_e ? n_e : 0, with_e ? o.e_nb : 0, ldx, ratio_m);
            }
          }
          if (rc) {
            hipDeviceSynchronize();
            return rc;
          }
        }
        for (const CopyBack &cb : copy_back)
          if (cb.g == g && lane_of_stratum(cb.y) == l) {
            const Stratum &t = c->st[cb.y];
            const size_t kq = (size_t)((t.K + 15) & ~15);
            hipLaunchKernelGGL(k_tables_home, dim3(32), dim3(256), 0, lanes[l], (const int32_t *)(b + o_halt) + cb.y, (int)cb.q,
                               (const double *)altC(cb.y), t.C, (size_t)t.K * c->D, (const double *)altCsq(cb.y), t.csq,
                               (const double *)altCnt(cb.y), t.counts, kq, (const double *)altBp(cb.y), t.Bpack, kq * c->KS * 4);
          }
        continue;
      }
      c->ahead_stat[5] += ny;
      const StepDev *dptr = d_steps + i;
      const bool contiguous = (int64_t)mine.size() == j - i;  // (a relaunch may run only some strata of a group)
      if (ranked[g] && contiguous && n_lanes == 1) {
        // the E-step of the whole group through the assignment kernels, then its outputs in the M-step's form, then the M-steps
        const hamsm_step &s0 = h_steps[i];
        rc = hamsm_assign_rows(c, (const char *)X + (size_t)s0.x_off * ldx * esz, dt, ldx, s0.n_span, (int32_t *)(b + o_lspan),
                               d_rows + s0.rows_off, (const unsigned long long *)(b + o_gcnt) + g * c->S);
        for (size_t c0_ = 0; c0_ < mine.size() && !rc; c0_ += PACK_NY) {
          const int nyc = (int)(mine.size() - c0_ < PACK_NY ? mine.size() - c0_ : PACK_NY);
          RowsPack rp;
          long long mnb = 0;
          int mK = 0;
          bool ap = false;
          for (int q = 0; q < nyc; q++) {
            const int64_t sq = mine[c0_ + q];
            const int y = h_steps[sq].stratum;
            pk.s[q] = sd[sq];
            int lo = 0;
            for (int y2 = 0; y2 < y; y2++) lo += c->st[y2].K;
            rp.off[q] = lo;
            ap = ap || sd[sq].perm;
            mnb = h_steps[sq].n_b > mnb ? h_steps[sq].n_b : mnb;
            mK = c->st[y].K > mK ? c->st[y].K : mK;
          }
          for (int q = nyc; q < PACK_NY; q++) rp.off[q] = 0;
          pk.n_next = 0;
          const dim3 gr((unsigned)((mnb + 63) / 64), (unsigned)nyc);
          if (dt == HAMSM_F32) {
            hipLaunchKernelGGL(k_rows_to_bits<float>, gr, dim3(256), 0, lanes[l], pk, rp, (const int32_t *)(b + o_lspan), ldx, c->D, c->KS, ratio);
            launch_update_only<float>(c, lanes[l], pk, nyc, mK, ldx, ratio, ap);
          } else {
            hipLaunchKernelGGL(k_rows_to_bits<double>, gr, dim3(256), 0, lanes[l], pk, rp, (const int32_t *)(b + o_lspan), ldx, c->D, c->KS, ratio);
            launch_update_only<double>(c, lanes[l], pk, nyc, mK, ldx, ratio, ap);
          }
        }
        if (rc) {
          hipDeviceSynchronize();
          return rc;
        }
        c->ranked_groups++;
        continue;
      }
      if (!contiguous && (ny > PACK_NY || generic)) {
        // only some strata of a group that travels by descriptors in memory: PACK_NY of them per launch, by value
        for (size_t c0_ = 0; c0_ < mine.size() && !rc; c0_ += PACK_NY) {
          const int nyc = (int)(mine.size() - c0_ < PACK_NY ? mine.size() - c0_ : PACK_NY);
          long long mnb = 0;
          int mK = 0;
          bool ap = false;
          for (int q = 0; q < nyc; q++) {
            const int64_t sq = mine[c0_ + q];
            pk.s[q] = sd[sq];
            ap = ap || sd[sq].perm;
            mnb = h_steps[sq].n_b > mnb ? h_steps[sq].n_b : mnb;
            mK = c->st[h_steps[sq].stratum].K > mK ? c->st[h_steps[sq].stratum].K : mK;
          }
          pk.n_next = 0;
          if (mnb <= 0) continue;
          rc = (dt == HAMSM_F32) ? launch_steps<float>(c, lanes[l], nullptr, pk, nyc, mnb, mK, ldx, ratio, ap)
                                 : launch_steps<double>(c, lanes[l], nullptr, pk, nyc, mnb, mK, ldx, ratio, ap);
        }
        if (rc) {
          hipDeviceSynchronize();
          return rc;
        }
        continue;
      }
      if (ny <= PACK_NY && !generic) {
        dptr = nullptr;  // descriptors by value
        for (int q = 0; q < ny; q++) pk.s[q] = sd[mine[q]];
        pk.n_next = 0;
        if (g + 2 < gstart.size()) {
          // the same strata of the next group, in the same positions
          int nn = 0;
          for (int64_t q = j; q < gstart[g + 2] && nn < ny; q++) {
            if (!runs(q) || lane_of(q) != l) continue;
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
  c->capturing = false;
  if (graph) {
    hipGraph_t gr = nullptr;
    hipGraphExec_t ge = nullptr;
    HIPCHK(c, hipStreamEndCapture(c->stream, &gr));
    clock_gettime(CLOCK_MONOTONIC, &tg1);
    HIPCHK(c, hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    clock_gettime(CLOCK_MONOTONIC, &tg2);
    hipEvent_t eg0 = nullptr, eg1 = nullptr;
    hipEventCreate(&eg0);
    hipEventCreate(&eg1);
    hipEventRecord(eg0, c->stream);
    HIPCHK(c, hipGraphLaunch(ge, c->stream));
    hipEventRecord(eg1, c->stream);
    clock_gettime(CLOCK_MONOTONIC, &tg3);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timespec tg4;
    clock_gettime(CLOCK_MONOTONIC, &tg4);
    float dev_ms = 0.f;
    hipEventElapsedTime(&dev_ms, eg0, eg1);
    hipEventDestroy(eg0);
    hipEventDestroy(eg1);
    if (dbg_enq) fprintf(stderr, "[hamsm] ... the graph on the device, event to event: %.2f ms\n", dev_ms);
    auto ms = [](const timespec &a, const timespec &b_) { return (b_.tv_sec - a.tv_sec) * 1e3 + (b_.tv_nsec - a.tv_nsec) * 1e-6; };
    if (dbg_enq)
      fprintf(stderr, "[hamsm] training pass as a hipGraph: capture %.2f ms, instantiate %.2f ms, launch call %.2f ms, device %.2f ms\n",
              ms(tg0, tg1), ms(tg1, tg2), ms(tg2, tg3), ms(tg3, tg4));
    hipGraphExecDestroy(ge);
    hipGraphDestroy(gr);
  }
  {
    // (a stratum's last step may have run inside a segment, its descriptor naming the alternate tables: the check reads the
    // stratum's own, where the segment's end has brought them)
    std::vector<StepDev> tails;
    for (int64_t i : tail_checks)
      if (runs(i)) {
        StepDev d = sd[i];
        const Stratum &t = c->st[h_steps[i].stratum];
        d.C = t.C, d.counts = t.counts, d.csq = t.csq, d.Bpack = t.Bpack;
        tails.push_back(d);
      }
    if (!tails.empty()) {
      HIPCHK(c, hipMemcpyAsync(b + o_tail, tails.data(), sizeof(StepDev) * tails.size(), hipMemcpyHostToDevice, c->stream));
      for (size_t k = 0; k < tails.size(); k++)
        hipLaunchKernelGGL(k_step_check, dim3(1), dim3(256), 0, c->stream, (const StepDev *)(b + o_tail) + k, ratio, (int)n_steps);
      HIPCHK(c, hipStreamSynchronize(c->stream));  // (`tails` is a local)
    }
  }
  hamsm_timer_end(c, ST_STEP);
  if (hipError_t le = hipGetLastError(); le != hipSuccess) {
    hipDeviceSynchronize();
    return hamsm_fail(c, HAMSM_EDEVICE, "step launches: %s", hipGetErrorString(le));
  }
  if (!h_report) break;  // (the stepwise entry point: one step, no stop protocol)
  // where every chain stands
  std::vector<int32_t> hw(2 * (size_t)c->S);
  {
    hipError_t e = hipMemcpyAsync(hw.data(), b + o_halt, sizeof(int32_t) * 2 * c->S, hipMemcpyDeviceToHost, c->stream);
    hipError_t es = hipStreamSynchronize(c->stream);  // always: `sd`, `ylist` and `hw` are locals the copies above read / write
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "minibatch_run read-back: %s", hipGetErrorString(e));
  }
  // a chain that stopped inside a pipelined segment: its tables are where its last applied step wrote them
  if (ahead_can)
    for (int s_ = 0; s_ < c->S; s_++) {
      if (!live[s_] || hw[s_] == 0x7f7f7f7f) continue;
      int64_t last = -1;
      for (int64_t q = (int64_t)hw[s_] - 1; q >= from[s_] && q >= 0; q--)
        if (h_steps[q].stratum == s_ && h_steps[q].n_b > 0) {
          last = q;
          break;
        }
      if (last >= 0 && ahd[last].pos >= 0 && out_alt[last]) {
        const Stratum &t = c->st[s_];
        const size_t kq = (size_t)((t.K + 15) & ~15);
        HIPCHK(c, hipMemcpyAsync(t.C, altC(s_), sizeof(double) * (size_t)t.K * c->D, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(t.csq, altCsq(s_), sizeof(double) * kq, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(t.counts, altCnt(s_), sizeof(double) * (size_t)t.K, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(t.Bpack, altBp(s_), sizeof(double) * kq * c->KS * 4, hipMemcpyDeviceToDevice, c->stream));
      }
    }
  bool again = false;
  for (int s_ = 0; s_ < c->S; s_++) {
    if (!live[s_]) continue;
    if (hw[c->S + s_] == HAMSM_STOP_AHEAD) {
      // a look-ahead block was given up at step hw[s_] (nothing of that step applied): relaunch this stratum from there
      from[s_] = hw[s_];
      again = true;
      ahead_fails++;
      c->ahead_stat[2]++;
      if (dbg_ahead) {
        const AheadDev &d = ahd[hw[s_]];
        double sb[2] = {0, 0}, seen = 0, sl[AHEAD_SLOTS];
        int32_t cn[4] = {0, 0, 0, 0};
        hipMemcpy(sb + 1, d.bound, sizeof(double), hipMemcpyDeviceToHost);
        hipMemcpy(cn, d.cnt, sizeof(cn), hipMemcpyDeviceToHost);
        if (d.prev_slot) {
          hipMemcpy(sl, d.prev_slot, sizeof(sl), hipMemcpyDeviceToHost);
          for (double v_ : sl) seen = (v_ > seen || v_ != v_) ? v_ : seen;
        }
        fprintf(stderr, "[hamsm] look-ahead block of stratum %d given up at step %d (pass %d): position %d, displacement seen %.6g, "
                "bound %.6g, undecided rows %d, open rows %d, unserved flags %d\n", s_, (int)hw[s_], pass, d.pos, seen, sb[1], cn[0], cn[2], cn[1]);
      }
    } else {
      fin_halt[s_] = hw[s_];
      fin_reason[s_] = hw[c->S + s_];
      live[s_] = 0;
    }
  }
  if (!again) break;
  c->ahead_stat[4]++;
  if (ahead_fails > 8 + (int)(n_steps / 128)) ahead_on = false;  // this data does not hold still: step by step from here
  // the given-up blocks left their later positions' bitmaps and tables behind
  HIPCHK(c, hipMemsetAsync(b + o_mb, 0, mb_bytes, c->stream));
  for (int s_ = 0; s_ < c->S; s_++)
    if (live[s_]) {
      const int32_t never = 0x7f7f7f7f, zero = 0;
      HIPCHK(c, hipMemcpyAsync(b + o_halt + sizeof(int32_t) * s_, &never, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(b + o_halt + sizeof(int32_t) * (c->S + s_), &zero, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
  HIPCHK(c, hipStreamSynchronize(c->stream));  // (`never` / `zero` are locals)
  }  // passes
  }  // !persist_done
  else if (h_report) {
    std::vector<int32_t> hw(2 * (size_t)c->S);
    HIPCHK(c, hipMemcpyAsync(hw.data(), b + o_halt, sizeof(int32_t) * 2 * c->S, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int s_ = 0; s_ < c->S; s_++) {
      fin_halt[s_] = hw[s_];
      fin_reason[s_] = hw[c->S + s_];
    }
  }
  if (h_report) {
    hipError_t e = hipMemcpyAsync(h_report, b + o_rep, sizeof(int32_t) * 2 * n_steps, hipMemcpyDeviceToHost, c->stream);
    hipError_t es = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "minibatch_run read-back: %s", hipGetErrorString(e));
    if (getenv("HAMSM_DEBUG_ENQUEUE")) {
      timespec ts_out;
      clock_gettime(CLOCK_MONOTONIC, &ts_out);
      fprintf(stderr, "[hamsm] run_steps: %.3f ms from entry to the results on the host\n",
              (ts_out.tv_sec - ts_in.tv_sec) * 1e3 + (ts_out.tv_nsec - ts_in.tv_nsec) * 1e-6);
    }
    if (h_resume && h_reason) {
      // stop word -> first step of the stratum that was not applied
      for (int s = 0; s < c->S; s++) {
        h_reason[s] = fin_reason[s];
        h_resume[s] = n_steps;
      }
      for (int64_t i = n_steps - 1; i >= 0; i--) {
        const int s = h_steps[i].stratum;
        if (i >= fin_halt[s]) h_resume[s] = i;
      }
    }
  } else {
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the descriptor upload reads from the local `sd`
  }
  return HAMSM_OK;
}

// (debug hooks, outside the ABI) the E-step of heavy training calls through the assignment kernels: mode 0 never, 1 by size, 2 always;
// returns the launch groups served that way since the last call (and resets the count)
extern "C" long long hamsm_debug_set_ranked(hamsm_ctx *c, int mode, double min_mflop) {
  if (hamsm_ctx_check(c, __func__)) return -1;
  if (mode >= 0) c->train_ranked = mode > 2 ? 2 : mode;
  if (min_mflop > 0) c->ranked_min_flop = min_mflop * 1e6;
  const long long n = c->ranked_groups;
  c->ranked_groups = 0;
  return n;
}
// (debug hooks, outside the ABI) look-ahead blocks: knobs and counters of a context
extern "C" int hamsm_debug_set_ahead(hamsm_ctx *c, int on, int n0, int gmax, double alpha) {
  CTX(c);
  c->ahead = on != 0;
  if (n0 >= 1) c->ahead_n0 = n0;
  if (gmax >= 1) c->ahead_gmax = gmax > 16 ? 16 : gmax;
  if (alpha >= 1.0) c->ahead_alpha = alpha;
  return HAMSM_OK;
}
// out[6]: blocks launched, steps inside them, blocks given up, (unused), relaunch passes, steps run step by step; reset != 0 zeroes
extern "C" int hamsm_debug_ahead_stats(hamsm_ctx *c, long long *out, int reset) {
  CTX(c);
  for (int i = 0; i < 6; i++) {
    if (out) out[i] = c->ahead_stat[i];
    if (reset) c->ahead_stat[i] = 0;
  }
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

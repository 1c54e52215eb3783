// flux_rows.hip -- deterministic flux accumulation with hand-written counting passes instead of a library sort (mode 1 of
// hamsm_flux_accumulate where the matrix has few enough rows; round 5).
//
// What has to come out, bit for bit (_fluxmatrix.py:143-146, 325-327; oracle hamsm_flux_accumulate): per WE iteration the
// COO -> dense conversion adds duplicate (start, end) pairs in segment order, and the iterations' matrices are added in ascending
// iteration order on top of what F holds.  flux_sorted.hip gets there with a stable radix sort of all segments by cell
// (rocPRIM) and four ordered passes over the whole array.  Here the order is made by two stable COUNTING passes that use what
// the sort cannot know -- the segments arrive grouped by iteration, and a cell is (row, column) with both below a few thousand:
//
//   k_fr_count   one workgroup per TILE (an iteration's segments cut into `split` equal parts): (row, column) of every segment
//                (basis / target overrides, range check), the tile's per-row counts
//   k_fr_scan_t  per row the running count over the tiles (where the row's segments of a tile start inside the row's list)
//   k_fr_scan_r  the rows' first positions (exclusive scan of the totals; one workgroup)
//   k_fr_place   level 1, one wave per tile, 64 segments at a time in order: segment -> its row's list (stable: rank among the
//                chunk's earlier segments of the same row + the row's running position); the common chunk -- 64 different rows
//                -- is recognised by one LDS round trip (every lane writes its lane number under its row and reads it back)
//   k_fr_row     level 2 and the sums, one workgroup per row of F, whose list is now in (iteration, segment) order:
//                  A  the waves take equal contiguous parts of the list and count their columns (LDS)
//                  B  columns' first positions (scan over waves, then over columns)
//                  C  every wave places its part, in order: the list is now in (column, iteration, segment) order
//                  D  every (column, iteration) run is summed in segment order by the thread at its first element; the run sums
//                     are packed, in order (a column's partials become contiguous)
//                  E  one thread per column adds its partials in ascending iteration on top of F[row][column]
//
// No atomics on F, no library call, no data-dependent order anywhere: the result does not depend on scheduling.
#include "common.h"

struct FluxArgs {
  const int32_t *pa, *ch;
  const double *p0, *p1, *w;
  int64_t i0, n;
  int K;
  double *F;
  int *err;
};

#define FR_INVALID 0xffffffffu

// tile -> its segments [s0, s1) and its iteration.  Iteration t is "the last t with offs[t] <= i" (as k_flux_keys): the first one
// reaches down to i0, the last one up to i0 + n, whatever offs says; everything is clamped into the call's range.
__device__ __forceinline__ void fr_tile(const FluxArgs &g, const int64_t *__restrict__ offs, int n_iters, int split, int tile,
                                        int64_t &s0, int64_t &s1, int &t) {
  t = tile / split;
  const int q = tile - t * split;
  const int64_t lo = g.i0, hi = g.i0 + g.n;
  int64_t a = (t == 0 || !offs) ? lo : offs[t], b = (t == n_iters - 1 || !offs) ? hi : offs[t + 1];
  a = a < lo ? lo : (a > hi ? hi : a);
  b = b < a ? a : (b > hi ? hi : b);
  const int64_t len = b - a;
  s0 = a + len * q / split;
  s1 = a + len * (q + 1) / split;
}

__global__ __launch_bounds__(256) void k_fr_count(MapperDev m, FluxArgs g, const int64_t *__restrict__ offs, int n_iters, int split,
                                                  int R, unsigned *__restrict__ rc, int32_t *__restrict__ cnt) {
  extern __shared__ unsigned fr_lh[];  // [R]
  const int tid = threadIdx.x;
  int64_t s0, s1;
  int t;
  fr_tile(g, offs, n_iters, split, blockIdx.x, s0, s1, t);
  for (int r = tid; r < R; r += 256) fr_lh[r] = 0;
  __syncthreads();
  const int K = g.K;
  for (int64_t i = s0 + tid; i < s1; i += 256) {
    double q0[HAMSM_MAX_NDIM], q1[HAMSM_MAX_NDIM];
    for (int d = 0; d < m.ndim; d++) {
      q0[d] = g.p0[i * m.ndim + d];
      q1[d] = g.p1[i * m.ndim + d];
    }
    int a = g.pa[i], b = g.ch[i];
    if (dev_in_box(m.target, m.ndim, q1)) b = K + 1;  // _fluxmatrix.py:135
    if (dev_in_box(m.basis, m.ndim, q0)) a = K;       // :136
    if (dev_in_box(m.basis, m.ndim, q1)) b = K;       // :137
    unsigned code = FR_INVALID;
    if (a < 0 || b < 0 || a >= R || b >= R) {
      dev_set_error(g.err, HAMSM_ERANGE, (int)i, a < 0 || a >= R ? a : b);
    } else {
      code = ((unsigned)a << 16) | (unsigned)b;
      atomicAdd(&fr_lh[a], 1u);  // (a count: the order of the increments does not matter)
    }
    rc[i - g.i0] = code;
  }
  __syncthreads();
  for (int r = tid; r < R; r += 256) cnt[(size_t)blockIdx.x * R + r] = (int32_t)fr_lh[r];
}

// cnt[tile][r] -> the number of row r's segments in the tiles before; the row's total.  16 rows x 64 stretches of tiles per
// workgroup: a stretch is summed, the 64 sums are scanned, the stretch is rewritten with its running count.
__global__ __launch_bounds__(1024) void k_fr_scan_t(int32_t *__restrict__ cnt, int NT, int R, int32_t *__restrict__ rowtot) {
  __shared__ int part[64][17];
  const int rr = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int r = blockIdx.x * 16 + rr;
  const int per = (NT + 63) / 64, a = tl * per < NT ? tl * per : NT, b = a + per < NT ? a + per : NT;
  int s = 0;
  if (r < R) {
    int t = a;
    for (; t + 8 <= b; t += 8) {  // (the loads do not depend on each other: eight in flight)
      int v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = cnt[(size_t)(t + q) * R + r];
#pragma unroll
      for (int q = 0; q < 8; q++) s += v[q];
    }
    for (; t < b; t++) s += cnt[(size_t)t * R + r];
  }
  part[tl][rr] = s;
  __syncthreads();
  if (tl == 0) {
    int run = 0;
    for (int j = 0; j < 64; j++) {
      const int v = part[j][rr];
      part[j][rr] = run;
      run += v;
    }
    if (r < R) rowtot[r] = run;
  }
  __syncthreads();
  if (r >= R) return;
  int run = part[tl][rr];
  int t = a;
  for (; t + 8 <= b; t += 8) {
    int v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = cnt[(size_t)(t + q) * R + r];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      cnt[(size_t)(t + q) * R + r] = run;
      run += v[q];
    }
  }
  for (; t < b; t++) {
    const int v = cnt[(size_t)t * R + r];
    cnt[(size_t)t * R + r] = run;
    run += v;
  }
}

// rowbase[r] = segments of the rows before r (one workgroup; R <= a few thousand).  rowbase[R] = the segments placed: the
// passes after this one run only if that is what the tiles were counted to hold (n less the out-of-range ones, already an error).
__global__ __launch_bounds__(256) void k_fr_scan_r(const int32_t *__restrict__ rowtot, int R, int32_t *__restrict__ rowbase, int64_t n,
                                                   int *err) {
  __shared__ int part[256];
  const int tid = threadIdx.x, per = (R + 255) / 256, a = tid * per < R ? tid * per : R, b = a + per < R ? a + per : R;
  int s = 0;
  for (int r = a; r < b; r++) s += rowtot[r];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int j = 0; j < 256; j++) {
      const int v = part[j];
      part[j] = run;
      run += v;
    }
    rowbase[R] = run;
    if ((int64_t)run > n) dev_set_error(err, HAMSM_EINVAL, R, run);  // iteration offsets that overlap: nothing is placed (k_fr_place, k_fr_row)
  }
  __syncthreads();
  int run = part[tid];
  for (int r = a; r < b; r++) {
    rowbase[r] = run;
    run += rowtot[r];
  }
}

// lanes of the wave that hold the same value of `v` (among the lanes of `among`): one ballot per bit
__device__ __forceinline__ unsigned long long fr_match(unsigned v, int bits, unsigned long long among) {
  unsigned long long mm = among;
  for (int bit = 0; bit < bits; bit++) {
    const bool one = (v >> bit) & 1u;
    const unsigned long long bal = __ballot(one);
    mm &= one ? bal : ~bal;
  }
  return mm;
}

// Lanes of a wave talk to each other through LDS words below (a wave's LDS accesses complete in order).  The compiler must not
// forward a lane's own store to its next load of the same word, nor keep a loaded value across another lane's store: a compiler-only
// fence where that matters.  (NOT volatile: every volatile access is followed by s_waitcnt vmcnt(0) lgkmcnt(0), which serialises the
// chunk loop on the round trip of its scattered global stores -- measured: 4 us per 64 segments.)
#define FR_FENCE() asm volatile("" ::: "memory")

#define FR_PF 4  // chunks requested ahead of the one being placed (a wave walks its tile alone: nothing else hides the latency)

__global__ __launch_bounds__(64) void k_fr_place(FluxArgs g, const int64_t *__restrict__ offs, int n_iters, int split, int R, int rbits,
                                                 const unsigned *__restrict__ rc, const int32_t *__restrict__ rowpos,
                                                 const int32_t *__restrict__ rowbase, unsigned *__restrict__ okey,
                                                 double *__restrict__ ow) {
  extern __shared__ int fr_run[];                             // [R] next position of row r's list | [R] lane tags (ushort)
  unsigned short *tag = (unsigned short *)(fr_run + R);
  int *run = fr_run;
  const int lane = threadIdx.x;
  // workgroups go to the 8 XCDs in turn: XCD x takes the x-th eighth of the tiles, so that the tiles which write into the same
  // cache lines of a row's list (neighbours in tile order) do so through the same L2
  const int NTl = n_iters * split, per8 = (NTl + 7) / 8, tile = (int)(blockIdx.x & 7) * per8 + (int)(blockIdx.x >> 3);
  if (tile >= NTl) return;
  int64_t s0, s1;
  int t;
  fr_tile(g, offs, n_iters, split, tile, s0, s1, t);
  if (s0 == s1 || (int64_t)rowbase[R] > g.n) return;
  for (int r = lane; r < R; r += 64) run[r] = rowbase[r] + rowpos[(size_t)tile * R + r];
  FR_FENCE();
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  unsigned code_q[FR_PF];
  double w_q[FR_PF];
  // (no branch around a global load or store below: the waits are then counted exactly -- s_waitcnt vmcnt(14), the chunk's loads
  // were issued four chunks ago -- instead of "everything outstanding"; lanes without a segment store into a dump slot past the end)
  auto fetch = [&](int64_t i0, unsigned &code, double &wv) {
    const int64_t i = i0 + lane;
    const bool in = i < s1;
    const int64_t ic = in ? i : s1 - 1;
    const unsigned c0 = rc[ic - g.i0];
    const double w0 = g.w[ic];
    code = in ? c0 : FR_INVALID;
    wv = w0;
  };
#pragma unroll
  for (int q = 0; q < FR_PF; q++) fetch(s0 + 64 * (int64_t)q, code_q[q], w_q[q]);
  for (int64_t i0 = s0; i0 < s1; i0 += 64 * FR_PF) {
#pragma unroll
    for (int q = 0; q < FR_PF; q++) {
      const unsigned code = code_q[q];
      const double wv = w_q[q];
      fetch(i0 + 64 * (int64_t)(q + FR_PF), code_q[q], w_q[q]);  // (the chunk FR_PF chunks ahead takes the slot)
      const bool valid = code != FR_INVALID;
      const unsigned r = valid ? code >> 16 : 0u, b = code & 0xffffu;
      // 64 different rows?  every lane leaves its number under its row; a lane that reads another's back shares its row
      if (valid) tag[r] = (unsigned short)lane;
      FR_FENCE();
      const bool lost = valid && tag[r] != (unsigned short)lane;
      int rank = 0, count = 1;
      if (__ballot(lost)) {  // (wave-uniform) exact ranks among the chunk's segments of the same row, in segment order
        const unsigned long long mm = fr_match(r, rbits, __ballot(valid));
        rank = __popcll(mm & lt);
        count = __popcll(mm);
      }
      const int base = run[r];
      FR_FENCE();
      if (valid && rank == 0) run[r] = base + count;  // (after every lane of the row has read `base`: LDS accesses of a wave are in order)
      const int64_t pos = valid ? (int64_t)(base + rank) : g.n + lane;
      okey[pos] = (b << 16) | (unsigned)t;
      ow[pos] = wv;
      FR_FENCE();
    }
  }
}

// one workgroup of NW waves per row of F; see the head of the file.  key = column << 16 | iteration.
//   ka / wa  the row's list in (iteration, segment) order (k_fr_place); dead after phase C: wa then holds the packed run sums
//   kb / wb  the row's list in (column, iteration, segment) order
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_fr_row(int R, int cbits, int64_t n, const int32_t *__restrict__ rowbase,
                                                    const unsigned *__restrict__ ka, double *__restrict__ wa,
                                                    unsigned *__restrict__ kb, double *__restrict__ wb, double *__restrict__ F) {
  constexpr int NT = NW * 64;
  extern __shared__ int fr_sm[];  // cnt [NW][R] | colstart [R] | cpos [R] | pcnt [R] | wsum [NW]
  int *cnt = fr_sm, *colstart = fr_sm + (size_t)NW * R, *cpos = colstart + R, *pcnt = cpos + R, *wsum = pcnt + R;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if ((int64_t)rowbase[R] > n) return;
  const int e0 = rowbase[r], e1 = rowbase[r + 1];
  if (e0 == e1) return;  // no segment starts in this state: the row of F is unchanged
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  // ---- A: the waves' column counts ----
  for (int i = tid; i < NW * R; i += NT) cnt[i] = 0;
  for (int i = tid; i < R; i += NT) pcnt[i] = 0;
  __syncthreads();
  const int nchunk = (e1 - e0 + 63) / 64, per = (nchunk + NW - 1) / NW;
  const int64_t a64 = (int64_t)e0 + (int64_t)w * per * 64;
  const int a = a64 < e1 ? (int)a64 : e1, b = (int64_t)a + (int64_t)per * 64 < e1 ? a + per * 64 : e1;
  int *mycnt = fr_sm + (size_t)w * R;
#pragma unroll 4
  for (int p = a; p < b; p += 64) {
    const int idx = p + lane;
    const unsigned key = ka[idx < b ? idx : b - 1];
    if (idx < b) atomicAdd(&mycnt[key >> 16], 1);
  }
  __syncthreads();
  // ---- B: per column the waves' first positions (relative to the column's), the columns' first positions ----
  for (int col = tid; col < R; col += NT) {
    int run = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; w2++) {
      const int v = cnt[w2 * R + col];
      cnt[w2 * R + col] = run;
      run += v;
    }
    colstart[col] = run;
  }
  __syncthreads();
  if (w == 0) {
    int carry = 0;
    for (int base = 0; base < R; base += 64) {
      const int col = base + lane;
      const int v = col < R ? colstart[col] : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      FR_FENCE();
      if (col < R) colstart[col] = carry + incl - v;
      carry += __shfl(incl, 63, 64);
    }
  }
  __syncthreads();
  // ---- C: place, in order ----
  {
    unsigned key_q[FR_PF];
    double w_q[FR_PF];
    auto fetch = [&](int p, unsigned &key, double &wv) {  // (branch-free, as in k_fr_place)
      const int idx = p + lane;
      const int ic = idx < b ? idx : b - 1;
      key = ka[ic];
      wv = wa[ic];
    };
    if (a < b) {
#pragma unroll
      for (int q = 0; q < FR_PF; q++) fetch(a + 64 * q, key_q[q], w_q[q]);
      for (int p0 = a; p0 < b; p0 += 64 * FR_PF) {
#pragma unroll
        for (int q = 0; q < FR_PF; q++) {
          const int p = p0 + 64 * q;
          const unsigned key = key_q[q];
          const double wv = w_q[q];
          fetch(p + 64 * FR_PF, key_q[q], w_q[q]);
          const bool valid = p + lane < b;
          const unsigned col = key >> 16;
          const unsigned long long mm = fr_match(col, cbits, __ballot(valid));
          const int rank = __popcll(mm & lt), count = __popcll(mm);
          const int base = cnt[w * R + col];
          FR_FENCE();
          if (valid && rank == 0) cnt[w * R + col] = base + count;  // (LDS accesses of a wave are in order: every lane has read `base`)
          const int64_t pos = valid ? (int64_t)(e0 + colstart[col] + base + rank) : n + lane;
          kb[pos] = key;
          wb[pos] = wv;
          FR_FENCE();
        }
      }
    }
  }
  __syncthreads();  // (also orders the workgroup's global stores above before its loads below)
  // ---- D: the (column, iteration) runs' in-order sums, packed in order into wa[e0 ..] ----
  double *cpart = wa + e0;
  int carry = 0;
  for (int base = e0; base < e1; base += NT * 4) {
    const int j0 = base + tid * 4;  // four consecutive elements per thread
    unsigned k[4];
    double v[4];
    const unsigned kprev = (j0 > e0 && j0 - 1 < e1) ? kb[j0 - 1] : FR_INVALID;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const bool in = j0 + q < e1;
      k[q] = in ? kb[j0 + q] : FR_INVALID;
      v[q] = in ? wb[j0 + q] : 0.0;
    }
    bool st[4];
    double ps[4];
    int nst = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      st[q] = j0 + q < e1 && k[q] != (q ? k[q - 1] : kprev);
      nst += st[q];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      double p = 0.0;
      if (st[q]) {
        bool open = true;  // the run goes on
#pragma unroll
        for (int q2 = q; q2 < 4; q2++) {
          open = open && k[q2] == k[q];
          if (open) p = p + v[q2];  // segment order inside the iteration
        }
        // ... beyond this thread's four: eight at a time, requested before any is looked at (a hot cell's run -- basis -> basis --
        // has dozens of segments per iteration; one dependent load per segment was 1.4 ms at C3)
        for (int e = j0 + 4; open && e < e1; e += 8) {
          unsigned k8[8];
          double v8[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int eu = e + u < e1 ? e + u : e1 - 1;
            k8[u] = kb[eu];
            v8[u] = wb[eu];
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            open = open && e + u < e1 && k8[u] == k[q];
            if (open) p = p + v8[u];
          }
        }
      }
      ps[q] = p;
    }
    int incl = nst;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int pre = 0, tot = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; w2++) {
      const int s = wsum[w2];
      pre += w2 < w ? s : 0;
      tot += s;
    }
    int pos = carry + pre + incl - nst;
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (st[q]) {
        const int col = (int)(k[q] >> 16);
        cpart[pos] = ps[q];
        atomicAdd(&pcnt[col], 1);                          // (a count)
        if (j0 + q - e0 == colstart[col]) cpos[col] = pos;  // the column's first run
        pos++;
      }
    carry += tot;
    __syncthreads();
  }
  // ---- E: per column, ascending iteration, on top of F ----
  for (int col = tid; col < R; col += NT) {
    const int np = pcnt[col];
    if (!np) continue;
    const double *pp = cpart + cpos[col];
    double acc = F[(size_t)r * R + col];
    int e = 0;
    for (; e + 8 <= np; e += 8) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = pp[e + q];
#pragma unroll
      for (int q = 0; q < 8; q++) acc = acc + v[q];
    }
    for (; e < np; e++) acc = acc + pp[e];
    F[(size_t)r * R + col] = acc;
  }
}

template <int NW>
static hipError_t fr_launch_row(hamsm_ctx *c, int R, int cbits, int64_t n, const int32_t *rowbase, const unsigned *ka, double *wa,
                                unsigned *kb, double *wb, double *F) {
  const size_t lds = sizeof(int) * ((size_t)NW * R + 3 * (size_t)R + NW);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)k_fr_row<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_fr_row<NW>, dim3((unsigned)R), dim3(NW * 64), lds, c->stream, R, cbits, n, rowbase, ka, wa, kb, wb, F);
  return hipSuccess;
}

// returns HAMSM_OK with *taken = 1 when the flux was accumulated here; *taken = 0: shape outside this path (the caller sorts)
int hamsm_flux_rows(hamsm_ctx *c, const MapperDev &m, const FluxArgs &g, const int64_t *d_offs, int64_t n_iters, int *taken) {
  *taken = 0;
  const int R = g.K + 2;
  const int64_t n = g.n;
  if (!d_offs) n_iters = 1;
  if (!c->flux_rows || n_iters < 1 || n_iters > 65536 || R > c->flux_rows_max_r || R > 65535 || n > 0x7fffffffLL || n <= 0) return HAMSM_OK;
  // tiles: an iteration in `split` parts of about 2048 segments, the count matrix [tiles][R] within 64 MB
  int64_t split = n / n_iters / 2048;
  split = split < 1 ? 1 : (split > 64 ? 64 : split);
  while (split > 1 && (size_t)n_iters * (size_t)split * (size_t)R * 4 > ((size_t)64 << 20)) split--;
  const int64_t NT = n_iters * split;
  if ((size_t)NT * (size_t)R > ((size_t)1 << 28)) return HAMSM_OK;
  const int NW = R <= 512 ? 16 : (R <= 2048 ? 8 : 4);
  if (sizeof(int) * ((size_t)NW * R + 3 * (size_t)R + NW) > 150 * 1024) return HAMSM_OK;
  int rbits = 1;
  while ((R - 1) >> rbits) rbits++;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return o;
  };
  // (rc, the segments' (row, column), is dead once they are placed: level 2 writes its keys there)
  // (+ 64: the dump slots of the lanes without a segment)
  const size_t o_rc = take(sizeof(unsigned) * ((size_t)n + 64)), o_cnt = take(sizeof(int32_t) * (size_t)NT * R),
               o_tot = take(sizeof(int32_t) * (size_t)R), o_base = take(sizeof(int32_t) * ((size_t)R + 1)),
               o_ka = take(sizeof(unsigned) * ((size_t)n + 64)), o_wa = take(sizeof(double) * ((size_t)n + 64)),
               o_wb = take(sizeof(double) * ((size_t)n + 64));
  int rc = hamsm_ensure_scratch(c, off);
  if (rc) return rc;
  char *b = (char *)c->scratch;
  unsigned *d_rc = (unsigned *)(b + o_rc), *d_ka = (unsigned *)(b + o_ka);
  int32_t *d_cnt = (int32_t *)(b + o_cnt), *d_tot = (int32_t *)(b + o_tot), *d_base = (int32_t *)(b + o_base);
  double *d_wa = (double *)(b + o_wa), *d_wb = (double *)(b + o_wb);
  hipLaunchKernelGGL(k_fr_count, dim3((unsigned)NT), dim3(256), sizeof(unsigned) * (size_t)R, c->stream, m, g, d_offs, (int)n_iters,
                     (int)split, R, d_rc, d_cnt);
  hipLaunchKernelGGL(k_fr_scan_t, dim3((unsigned)((R + 15) / 16)), dim3(1024), 0, c->stream, d_cnt, (int)NT, R, d_tot);
  hipLaunchKernelGGL(k_fr_scan_r, dim3(1), dim3(256), 0, c->stream, (const int32_t *)d_tot, R, d_base, n, g.err);
  hipLaunchKernelGGL(k_fr_place, dim3((unsigned)(((NT + 7) / 8) * 8)), dim3(64), sizeof(int) * (size_t)R + sizeof(unsigned short) * (size_t)R, c->stream, g,
                     d_offs, (int)n_iters, (int)split, R, rbits, (const unsigned *)d_rc, (const int32_t *)d_cnt, (const int32_t *)d_base,
                     d_ka, d_wa);
  hipError_t e = NW == 16 ? fr_launch_row<16>(c, R, rbits, n, d_base, d_ka, d_wa, d_rc, d_wb, g.F)
                 : NW == 8 ? fr_launch_row<8>(c, R, rbits, n, d_base, d_ka, d_wa, d_rc, d_wb, g.F)
                           : fr_launch_row<4>(c, R, rbits, n, d_base, d_ka, d_wa, d_rc, d_wb, g.F);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "flux_rows: %s", hipGetErrorString(e));
  *taken = 1;
  return HAMSM_OK;
}

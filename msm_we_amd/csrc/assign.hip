// assign.hip -- a3/a4 classify and a9 stratified assignment (StratifiedClusters.predict, reference
// msm_we/stratified_clustering.py:101-212) for gfx950.
//
// Kernel plan.
//   k_sort_count / k_sort_offsets / k_sort_fill : three light passes bucket the frames by stratum
//       s = remap[bin(pcoord)] into a permutation (4 B per frame); basis / target frames receive their label
//       here and never reach the distance kernel.
//   k_assign_sorted : every workgroup owns <= 512 frames of ONE stratum.  The stratum's centroids, pre-packed as
//       MFMA fragments, are loaded into registers once per workgroup (each wave keeps JW centroid tiles); the
//       rows are gathered through the permutation into LDS (each 4*D-byte row is one contiguous read); the waves
//       sweep the 16-frame tiles with v_mfma_f64_16x16x4_f64, centroids as the A operand (output rows) and
//       frames as the B operand (output columns), so that a lane holds, for ONE frame, its distances to the
//       centroids (l>>4)+4r of each tile: csq - 2*dot and the running first-minimum are register-local, only the
//       four lane groups and then the waves are merged (2 shuffles + one LDS round per batch of tiles).
// The dot product of a frame and a centroid is an f64 FMA chain over ascending k (the MFMA applies its four
// k-products in order; zero padding adds exact zeros), i.e. the arithmetic contract of include/hamsm.h;
// hamsm_selftest_mfma() checks that property of the hardware bit for bit.
//
// Measured on MI355X (C3: 10 M rows x 30 features, 4 strata x 250 centroids): 3.5 ms = 42 TFLOP/s; earlier
// forms of this kernel are in the history (chunk-local buckets with L2-streamed fragments: 7.1 ms; 16-lane DPP
// argmin: 4.8 ms) -- see DESIGN.md §5.1 for what each step bought.
#include "common.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef const double __attribute__((address_space(1))) *gcdp;  // global-address-space view: global_load, not flat_load

#define NTHREADS 256

struct AssignArgs {
  const void *X;
  int64_t ldx;
  const double *pcoord;
  int64_t n;
  int32_t *labels;
  int32_t *seen;  // nullable [2][S]
  const StratumDev *table;
  int S, D, Kfit;
  int *err;
};

// ---- classify -----------------------------------------------------------------------------------
__global__ void k_classify(MapperDev m, const double *__restrict__ p, int64_t n, int apply_remap, int32_t *bin,
                           uint8_t *flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double pi[HAMSM_MAX_NDIM];
  for (int d = 0; d < m.ndim; d++) pi[d] = p[i * m.ndim + d];
  int b = dev_bin_of(m, pi);
  if (b >= 0 && apply_remap) b = m.remap[b];
  if (bin) bin[i] = b;
  if (flags) flags[i] = (uint8_t)((dev_in_box(m.basis, m.ndim, pi) ? 1 : 0) | (dev_in_box(m.target, m.ndim, pi) ? 2 : 0));
}

extern "C" int hamsm_classify(hamsm_ctx *c, const double *p, int64_t n, int apply_remap, int32_t *bin, uint8_t *flags) {
  if (!c->have_mapper || !c->have_bounds) return hamsm_fail(c, HAMSM_EINVAL, "mapper/bounds not set");
  if (n <= 0) return HAMSM_OK;
  hipLaunchKernelGGL(k_classify, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->mapper, p, n, apply_remap,
                     bin, flags);
  HIPCHK(c, hipGetLastError());
  return HAMSM_OK;
}

#ifdef ASSIGN_PROFILE
__device__ unsigned long long g_assign_prof[8];
#define APROF_INIT unsigned long long t_last_ = __builtin_amdgcn_s_memtime(), t_acc_[6] = {0, 0, 0, 0, 0, 0};
#define APROF(i)                                             \
  {                                                          \
    unsigned long long t_ = __builtin_amdgcn_s_memtime();    \
    t_acc_[i] += t_ - t_last_;                               \
    t_last_ = t_;                                            \
  }
#define APROF_END                                                                        \
  if (threadIdx.x == 0) {                                                                \
    for (int i_ = 0; i_ < 6; i_++) atomicAdd(&g_assign_prof[i_], t_acc_[i_]);            \
    atomicAdd(&g_assign_prof[7], 1ull);                                                  \
  }
extern "C" int hamsm_debug_assign_profile(unsigned long long *h_out) {
  return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_assign_prof), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#else
#define APROF_INIT
#define APROF(i)
#define APROF_END
#endif

// ---- stratum-sorted assignment -------------------------------------------------------------------
#define FRB_MAX 512
struct SortArgs {
  const void *X;
  int64_t ldx;
  const double *pcoord;
  int64_t n;
  int32_t *labels;
  int32_t *seen;
  const StratumDev *table;
  int S, D, Kfit;
  int *err;
  int32_t *code;                 // [n] stratum of frame, or -1 (basis / target / error: label already final)
  int32_t *perm;                 // [n] frames grouped by stratum
  unsigned long long *cnt;       // [S] frames per stratum
  unsigned long long *cursor;    // [S] fill cursors
  long long *base;               // [S+1] first position of stratum s in perm
  long long *blkbase;            // [S+1] first workgroup of stratum s in the main kernel's grid
  int need_rb;
  int FRB;  // frames per workgroup of the main kernel (multiple of 16)
};

__global__ __launch_bounds__(256) void k_sort_count(MapperDev mp, SortArgs g) {
  extern __shared__ unsigned int lh[];  // [S]
  for (int s = threadIdx.x; s < g.S; s += 256) lh[s] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < g.n; i += (int64_t)gridDim.x * 256) {
    double pi[HAMSM_MAX_NDIM];
    for (int d = 0; d < mp.ndim; d++) pi[d] = g.pcoord[i * mp.ndim + d];
    int b = dev_bin_of(mp, pi);
    int code = -1;
    if (b < 0) {
      dev_set_error(g.err, HAMSM_ERANGE, (int)i, b);
    } else {
      int s = mp.remap[b];                       // stratified_clustering.py:134-135
      if (dev_in_box(mp.target, mp.ndim, pi)) {  // :159 target first
        g.labels[i] = g.Kfit + 1;
        if (g.seen) g.seen[s] = 1;
      } else if (dev_in_box(mp.basis, mp.ndim, pi)) {  // :165
        g.labels[i] = g.Kfit;
        if (g.seen) g.seen[g.S + s] = 1;
      } else if (g.table[s].K == 0) {  // :187-189
        dev_set_error(g.err, HAMSM_EUNFIT, (int)i, s);
      } else {
        code = s;
        atomicAdd(&lh[s], 1u);
      }
    }
    g.code[i] = code;
  }
  __syncthreads();
  for (int s = threadIdx.x; s < g.S; s += 256)
    if (lh[s]) atomicAdd(&g.cnt[s], (unsigned long long)lh[s]);
}

__global__ void k_sort_offsets(SortArgs g) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    long long b = 0, k = 0;
    for (int s = 0; s < g.S; s++) {
      g.base[s] = b;
      g.blkbase[s] = k;
      b += (long long)g.cnt[s];
      k += ((long long)g.cnt[s] + g.FRB - 1) / g.FRB;
      g.cursor[s] = 0;
    }
    g.base[g.S] = b;
    g.blkbase[g.S] = k;
  }
}

__global__ __launch_bounds__(256) void k_sort_fill(SortArgs g) {
  extern __shared__ unsigned int lh[];  // [S] local counts, then local cursors
  unsigned long long *lbase = (unsigned long long *)(lh + ((g.S + 1) & ~1));  // [S] reserved range of this workgroup
  const int64_t per = (g.n + gridDim.x - 1) / gridDim.x, a = blockIdx.x * per, b = (a + per < g.n) ? a + per : g.n;
  for (int s = threadIdx.x; s < g.S; s += 256) lh[s] = 0;
  __syncthreads();
  for (int64_t i = a + threadIdx.x; i < b; i += 256) {
    int c = g.code[i];
    if (c >= 0) atomicAdd(&lh[c], 1u);
  }
  __syncthreads();
  for (int s = threadIdx.x; s < g.S; s += 256) {
    lbase[s] = lh[s] ? atomicAdd(&g.cursor[s], (unsigned long long)lh[s]) : 0ull;
    lh[s] = 0;
  }
  __syncthreads();
  for (int64_t i = a + threadIdx.x; i < b; i += 256) {
    int c = g.code[i];
    if (c >= 0) {
      unsigned int r = atomicAdd(&lh[c], 1u);
      g.perm[g.base[c] + (long long)lbase[c] + r] = (int32_t)i;
    }
  }
}

// NW waves per workgroup share the staged rows; each wave owns JW = 16/NW... centroid tiles of the current group
// (NW*JW tiles per group).  8 waves x 2 tiles needs half the B registers of 4 x 4, so that two workgroups per CU
// give 4 waves per SIMD -- a single wave cannot keep the f64 MFMA pipe busy.
template <typename XT, int KS, int MT, int NW, int JW>
__global__ __launch_bounds__(NW * 64, NW / 2) void k_assign_sorted(SortArgs g) {
  APROF_INIT
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT = NW * 64;
  const int FRB = g.FRB;
  // which stratum does this workgroup serve?
  int s = 0;
  {
    const long long blk = blockIdx.x;
    if (blk >= g.blkbase[g.S]) return;
    while (g.blkbase[s + 1] <= blk) s++;
  }
  const long long p0 = g.base[s] + (long long)(blockIdx.x - g.blkbase[s]) * FRB;
  const long long pend = g.base[s + 1];
  const int nf = (int)((pend - p0) < FRB ? (pend - p0) : FRB);
  const int nts = (nf + 15) >> 4;
  XT *Xs = (XT *)smem;                                                   // [FRB][D]
  size_t xbytes = ((size_t)FRB * g.D * sizeof(XT) + 15) & ~(size_t)15;
  int *fidx = (int *)(smem + xbytes);                                    // [FRB] frame index of local row
  double *part_d = (double *)(fidx + FRB);                               // [2][NW][MT*16]
  int *part_j = (int *)(part_d + 2 * NW * MT * 16);                       // [2][NW][MT*16]
  double *rb_d = (double *)(part_j + 2 * NW * MT * 16);                   // [FRB] running best (need_rb)
  int *rb_j = (int *)(rb_d + (g.need_rb ? FRB : 0));
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double inf = __builtin_huge_val();
  const StratumDev sd = g.table[s];
  const int njt = (sd.K + 15) >> 4;
  const int ngrp = (njt + NW * JW - 1) / (NW * JW);

  for (int i = tid; i < nf; i += NT) fidx[i] = g.perm[p0 + i];
  __syncthreads();
  {
    const XT *src = (const XT *)g.X;
    const int tot = nf * g.D;
    // gather through the permutation, 8 independent loads in flight per thread (a plain loop would expose one
    // HBM round trip per element)
    constexpr int U = 8;
    typedef XT xt2 __attribute__((ext_vector_type(2)));
    if ((g.D & 1) == 0 && (g.ldx & 1) == 0 && ((uintptr_t)src & (2 * sizeof(XT) - 1)) == 0) {
      // rows start on 2-element boundaries: move pairs (half the load instructions, 16 elements in flight)
      const int D2 = g.D >> 1, tot2 = nf * D2;
      xt2 *Xs2 = (xt2 *)Xs;
      for (int e0 = tid; e0 < tot2; e0 += NT * U) {
        xt2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          int e = e0 + u * NT;
          int ec = e < tot2 ? e : 0;
          int r = ec / D2, k = ec - r * D2;
          v[u] = *(const xt2 *)(src + (int64_t)fidx[r] * g.ldx + 2 * k);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          int e = e0 + u * NT;
          if (e < tot2) Xs2[e] = v[u];
        }
      }
    } else
    for (int e0 = tid; e0 < tot; e0 += NT * U) {
      XT v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        int e = e0 + u * NT;
        int ec = e < tot ? e : 0;
        int r = ec / g.D, k = ec - r * g.D;
        v[u] = src[(int64_t)fidx[r] * g.ldx + k];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        int e = e0 + u * NT;
        if (e < tot) Xs[e] = v[u];
      }
    }
  }
  __syncthreads();
  APROF(0)
  int batch = 0;
  for (int gi = 0; gi < ngrp; gi++) {
    // centroids are the MFMA's A operand (output rows), frames its B operand (output columns): lane l then holds,
    // for frame l&15 of the tile, the dot products with centroids (l>>4) + 4r -- the first-minimum over a tile's
    // centroids is a register-only reduction, and only the four lane groups have to be merged across lanes
    double b[JW][KS], cs[JW][4];
#pragma unroll
    for (int u = 0; u < JW; u++) {
      const int jt = gi * NW * JW + wv * JW + u;
      const bool live = jt < njt;
      const int jts = live ? jt : 0;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        double c0 = ((gcdp)sd.csq)[jts * 16 + (lane >> 4) + 4 * r];
        cs[u][r] = live ? c0 : inf;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        double v = ((gcdp)sd.Bpack)[((size_t)jts * KS + ks) * 64 + lane];
        b[u][ks] = live ? v : 0.0;
      }
    }
    APROF(1)
    for (int t = 0; t < nts; t += MT, batch++) {
      const int nt = (nts - t) < MT ? (nts - t) : MT;
      double bd[MT];
      int bj[MT];
#pragma unroll
      for (int m = 0; m < MT; m++) {
        bd[m] = inf;
        bj[m] = 0x7fffffff;
        if (m >= nt) continue;  // block-uniform
        double a[KS];
        {
          const int lr = (t + m) * 16 + (lane & 15);
          const bool ok = lr < nf;
          const XT *xr = Xs + (ok ? lr : 0) * g.D;
#pragma unroll
          for (int ks = 0; ks < KS; ks++) {
            int k = ks * 4 + (lane >> 4);
            double v = (double)xr[k < g.D ? k : 0];
            a[ks] = (ok && k < g.D) ? v : 0.0;
          }
        }
        v4f64 acc[JW];
#pragma unroll
        for (int u = 0; u < JW; u++) acc[u] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
          for (int u = 0; u < JW; u++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[u][ks], a[ks], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < JW; u++) {
          const int j0 = (gi * NW * JW + wv * JW + u) * 16 + (lane >> 4);
#pragma unroll
          for (int r = 0; r < 4; r++) {  // centroid j0 + 4r: ascending, so strict < keeps the lowest index
            double d = __builtin_fma(-2.0, acc[u][r], cs[u][r]);
            bool lt = d < bd[m];
            bd[m] = lt ? d : bd[m];
            bj[m] = lt ? j0 + 4 * r : bj[m];
          }
        }
      }
      APROF(2)
      const int buf = batch & 1;
#pragma unroll
      for (int m = 0; m < MT; m++) {
        // merge the four lane groups (lanes l, l^16, l^32, l^48 hold the same frame)
        double d = bd[m];
        int j = bj[m];
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
          double d2 = __shfl_xor(d, off, 64);
          int j2 = __shfl_xor(j, off, 64);
          bool take = (d2 < d) || (d2 == d && j2 < j);
          d = take ? d2 : d;
          j = take ? j2 : j;
        }
        if (lane < 16 && m < nt) {
          part_d[(buf * NW + wv) * MT * 16 + m * 16 + lane] = d;
          part_j[(buf * NW + wv) * MT * 16 + m * 16 + lane] = j;
        }
      }
      APROF(3)
      __syncthreads();
      APROF(4)
      if (tid < nt * 16) {
        const int m = tid >> 4, row = tid & 15;
        double d = part_d[(buf * NW + 0) * MT * 16 + m * 16 + row];
        int j = part_j[(buf * NW + 0) * MT * 16 + m * 16 + row];
#pragma unroll
        for (int w = 1; w < NW; w++) {
          double d2 = part_d[(buf * NW + w) * MT * 16 + m * 16 + row];
          int j2 = part_j[(buf * NW + w) * MT * 16 + m * 16 + row];
          bool take = (d2 < d) || (d2 == d && j2 < j);
          d = take ? d2 : d;
          j = take ? j2 : j;
        }
        const int lr = (t + m) * 16 + row;
        if (ngrp > 1) {
          if (gi > 0) {
            double d0 = rb_d[lr];
            int j0 = rb_j[lr];
            bool keep = (d0 < d) || (d0 == d && j0 < j);
            d = keep ? d0 : d;
            j = keep ? j0 : j;
          }
          if (gi < ngrp - 1) {
            rb_d[lr] = d;
            rb_j[lr] = j;
          }
        }
        if (gi == ngrp - 1 && lr < nf) {
          if (j == 0x7fffffff) j = 0;
          g.labels[fidx[lr]] = sd.off + j;  // stratified_clustering.py:195
        }
      }
      APROF(5)
    }
  }
  APROF_END
}

int hamsm_ensure_scratch2(hamsm_ctx *c, size_t bytes);

template <typename XT, int KS, int NW, int JW>
static int launch_sorted_cfg(hamsm_ctx *c, SortArgs g) {
  constexpr int MT = (KS <= 8) ? 4 : (KS <= 16 ? 2 : 1);
  g.need_rb = 0;
  for (int s = 0; s < c->S; s++) g.need_rb |= ((c->st[s].K + 15) / 16 > NW * JW);
  const int FRB = g.FRB;
  size_t xbytes = ((size_t)FRB * g.D * sizeof(XT) + 15) & ~(size_t)15;
  size_t lds = xbytes + sizeof(int) * FRB + (sizeof(double) + sizeof(int)) * 2 * NW * MT * 16 +
               (g.need_rb ? (sizeof(double) + sizeof(int)) * FRB : 0) + 16;
  if (lds > 160 * 1024) return hamsm_fail(c, HAMSM_ENOTSUP, "sorted assign needs %zu B of LDS", lds);
  auto kern = k_assign_sorted<XT, KS, MT, NW, JW>;
  HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  unsigned grid = (unsigned)((g.n + FRB - 1) / FRB + c->S);  // upper bound; surplus workgroups exit at once
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, c->stream, g);
  HIPCHK(c, hipGetLastError());
  return HAMSM_OK;
}

template <typename XT, int KS>
static int launch_sorted_ks(hamsm_ctx *c, SortArgs g) {
  // 8 waves x 2 centroid tiles while the k-steps are few (<= 128 VGPRs: 4 waves per SIMD); else 4 waves x (4 or 2)
  if constexpr (KS <= 8) {
    if (c->assign_waves == 8) return launch_sorted_cfg<XT, KS, 8, 2>(c, g);
    return launch_sorted_cfg<XT, KS, 4, 4>(c, g);
  } else if constexpr (KS <= 16) {
    return launch_sorted_cfg<XT, KS, 4, 4>(c, g);
  } else {
    return launch_sorted_cfg<XT, KS, 4, 2>(c, g);
  }
}

template <typename XT>
static int launch_sorted(hamsm_ctx *c, const AssignArgs &a) {
  const int64_t n = a.n;
  const int S = c->S;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return o;
  };
  size_t o_code = take(sizeof(int32_t) * n), o_perm = take(sizeof(int32_t) * n), o_cnt = take(sizeof(long long) * S),
         o_cur = take(sizeof(long long) * S), o_base = take(sizeof(long long) * (S + 1)), o_blk = take(sizeof(long long) * (S + 1));
  int rc = hamsm_ensure_scratch2(c, off);
  if (rc) return rc;
  char *b = (char *)c->scratch2;
  SortArgs g{a.X, a.ldx, a.pcoord, n, a.labels, a.seen, a.table, S, a.D, a.Kfit, a.err, (int32_t *)(b + o_code),
             (int32_t *)(b + o_perm), (unsigned long long *)(b + o_cnt), (unsigned long long *)(b + o_cur),
             (long long *)(b + o_base), (long long *)(b + o_blk), 0, FRB_MAX};
  // frames per workgroup: as many as keep the staged rows within ~64 KB (2 workgroups per CU)
  while (g.FRB > 64 && (size_t)g.FRB * a.D * sizeof(XT) > 64 * 1024) g.FRB >>= 1;
  HIPCHK(c, hipMemsetAsync(b + o_cnt, 0, sizeof(long long) * S, c->stream));
  unsigned gridc = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_sort_count, dim3(gridc), dim3(256), sizeof(unsigned int) * S, c->stream, c->mapper, g);
  hipLaunchKernelGGL(k_sort_offsets, dim3(1), dim3(64), 0, c->stream, g);
  hipLaunchKernelGGL(k_sort_fill, dim3(gridc), dim3(256), sizeof(unsigned int) * ((S + 1) & ~1) + sizeof(long long) * S,
                     c->stream, g);
  HIPCHK(c, hipGetLastError());
  hamsm_timer_end(c, ST_OTHER);   // the three bucket passes
  hamsm_timer_begin(c);           // ... and the distance kernel proper (ST_ASSIGN, ended by the caller)
  switch (c->KS) {
#define CASE(KSV) \
  case KSV: return launch_sorted_ks<XT, KSV>(c, g);
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(6) CASE(8) CASE(12) CASE(16) CASE(24) CASE(32)
#undef CASE
  }
  return hamsm_fail(c, HAMSM_ENOTSUP, "KS=%d", c->KS);
}

extern "C" int hamsm_assign(hamsm_ctx *c, const void *X, int dt, int64_t ldx, const double *p, int64_t n, int32_t *labels,
                            int32_t *seen) {
  if (!c->have_mapper || !c->have_bounds) return hamsm_fail(c, HAMSM_EINVAL, "mapper/bounds not set");
  if (dt != HAMSM_F32 && dt != HAMSM_F64) return hamsm_fail(c, HAMSM_EINVAL, "x_dtype");
  if (ldx < c->D) return hamsm_fail(c, HAMSM_EINVAL, "ldx < D");
  if (n <= 0) return HAMSM_OK;
  int rc = hamsm_refresh_table(c);
  if (rc) return rc;
  AssignArgs g{X, ldx, p, n, labels, seen, c->d_table, c->S, c->D, c->Kfit, c->d_err};
  hamsm_timer_begin(c);  // bucket passes -> ST_OTHER (ended inside launch_sorted), distance kernel -> ST_ASSIGN
  rc = (dt == HAMSM_F32) ? launch_sorted<float>(c, g) : launch_sorted<double>(c, g);
  hamsm_timer_end(c, ST_ASSIGN);
  if (rc) return rc;
  return hamsm_check_device_error(c, "assign");
}

extern "C" int hamsm_predict_host(hamsm_ctx *c, const void *X, int dt, const double *p, int64_t n, int32_t *labels,
                                  int32_t *seen) {
  if (n <= 0) return HAMSM_OK;
  size_t xs = (dt == HAMSM_F32 ? 4 : 8) * (size_t)n * c->D;
  size_t ps = sizeof(double) * (size_t)n * c->ndim, ls = sizeof(int32_t) * (size_t)n, ss = sizeof(int32_t) * 2 * c->S;
  size_t o_p = (xs + 255) & ~(size_t)255, o_l = (o_p + ps + 255) & ~(size_t)255, o_s = (o_l + ls + 255) & ~(size_t)255;
  int rc = hamsm_ensure_scratch(c, o_s + ss);
  if (rc) return rc;
  char *base = (char *)c->scratch;
  HIPCHK(c, hipMemcpyAsync(base, X, xs, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(base + o_p, p, ps, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(base + o_s, 0, ss, c->stream));
  rc = hamsm_assign(c, base, dt, c->D, (const double *)(base + o_p), n, (int32_t *)(base + o_l),
                    seen ? (int32_t *)(base + o_s) : nullptr);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(labels, base + o_l, ls, hipMemcpyDeviceToHost, c->stream));
  if (seen) HIPCHK(c, hipMemcpyAsync(seen, base + o_s, ss, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HAMSM_OK;
}

// ---- hardware property check: is v_mfma_f64_16x16x4_f64 an in-order FMA chain? --------------------
__global__ void k_selftest(const double *A, const double *B, int KS, double *out_mfma, double *out_chain) {
  // A [16][KS*4], B [KS*4][16] row-major; one wave
  int lane = threadIdx.x;
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  for (int ks = 0; ks < KS; ks++) {
    double a = A[(lane & 15) * KS * 4 + ks * 4 + (lane >> 4)];
    double b = B[(ks * 4 + (lane >> 4)) * 16 + (lane & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; r++) {
    int row = (lane >> 4) + 4 * r, col = lane & 15;
    out_mfma[row * 16 + col] = acc[r];
    double ch = 0.0;
    for (int k = 0; k < KS * 4; k++) ch = __builtin_fma(A[row * KS * 4 + k], B[k * 16 + col], ch);
    out_chain[row * 16 + col] = ch;
  }
}

// Not part of the product ABI (debug / test hook): returns the number of the 256 outputs whose bits
// differ between the MFMA and the FMA chain, for A (16 x 4*KS) and B (4*KS x 16) given on the host.
extern "C" int hamsm_selftest_mfma(hamsm_ctx *c, const double *hA, const double *hB, int KS, double *h_mfma,
                                   double *h_chain) {
  size_t na = sizeof(double) * 16 * KS * 4;
  int rc = hamsm_ensure_scratch(c, 2 * na + 2 * 256 * sizeof(double));
  if (rc) return rc;
  double *dA = (double *)c->scratch, *dB = dA + 16 * KS * 4, *o1 = dB + 16 * KS * 4, *o2 = o1 + 256;
  HIPCHK(c, hipMemcpyAsync(dA, hA, na, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dB, hB, na, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, c->stream, dA, dB, KS, o1, o2);
  HIPCHK(c, hipMemcpyAsync(h_mfma, o1, 256 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(h_chain, o2, 256 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int bad = 0;
  for (int i = 0; i < 256; i++) bad += memcmp(&h_mfma[i], &h_chain[i], 8) != 0;
  return bad;
}

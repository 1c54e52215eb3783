// flux.hip -- a10 flux-matrix accumulation (FluxMatrixMixin.build_flux_matrix, reference
// msm_we/_hamsm/_fluxmatrix.py:97-164, summed over iterations :325-327), a11 cluster pcoord
// statistics (_clustering.py:1559-1591) and the a13 feed: row-normalised T as CSR (_analysis.py:44-79).
//
// These are HBM/L2-atomic bound scatter kernels: 32 B read per segment, one f64
// atomic into a (K+2)^2 matrix that lives in L2 / Infinity Cache (8 MB at K=1000).
#include "common.h"

struct FluxArgs {
  const int32_t *pa, *ch;
  const double *p0, *p1, *w;
  int64_t i0, n;
  int K;
  double *F;
  int *err;
};

__global__ void k_flux_atomic(MapperDev m, FluxArgs g) {
  int64_t i = g.i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.i0 + g.n) return;
  double q0[HAMSM_MAX_NDIM], q1[HAMSM_MAX_NDIM];
  for (int d = 0; d < m.ndim; d++) {
    q0[d] = g.p0[i * m.ndim + d];
    q1[d] = g.p1[i * m.ndim + d];
  }
  int a = g.pa[i], b = g.ch[i];
  const int K = g.K;
  if (dev_in_box(m.target, m.ndim, q1)) b = K + 1;  // _fluxmatrix.py:135
  if (dev_in_box(m.basis, m.ndim, q0)) a = K;       // :136
  if (dev_in_box(m.basis, m.ndim, q1)) b = K;       // :137
  if (a < 0 || b < 0 || a >= K + 2 || b >= K + 2) {
    dev_set_error(g.err, HAMSM_ERANGE, (int)i, a < 0 || a >= K + 2 ? a : b);
    return;
  }
  // global_atomic_add_f64 (built with -munsafe-fp-atomics: no CAS loop)
  unsafeAtomicAdd(&g.F[(size_t)a * (K + 2) + b], g.w[i]);
}

// Device-scope f64 atomics are executed beyond the XCDs' L2s (the 8 L2s are not coherent with each other inside a kernel):
// 10^7 of them take 0.93 ms whatever the matrix size.  When 8 copies of the matrix are affordable, every XCD adds into
// ITS OWN copy with workgroup-scope atomics -- performed in its L2, at L2 rate -- and a second kernel sums the copies in
// XCD order into F.  (The XCD is read from the hardware register, not inferred from the workgroup id.)
#define FLUX_NXCD 8
#define FLUX_PART_MAX_BYTES ((size_t)512 << 20)
__global__ void k_flux_xcd(MapperDev m, FluxArgs g, double *__restrict__ part, size_t cells) {
  int64_t i = g.i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.i0 + g.n) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= FLUX_NXCD - 1;
  double q0[HAMSM_MAX_NDIM], q1[HAMSM_MAX_NDIM];
  for (int d = 0; d < m.ndim; d++) {
    q0[d] = g.p0[i * m.ndim + d];
    q1[d] = g.p1[i * m.ndim + d];
  }
  int a = g.pa[i], b = g.ch[i];
  const int K = g.K;
  if (dev_in_box(m.target, m.ndim, q1)) b = K + 1;  // _fluxmatrix.py:135
  if (dev_in_box(m.basis, m.ndim, q0)) a = K;       // :136
  if (dev_in_box(m.basis, m.ndim, q1)) b = K;       // :137
  if (a < 0 || b < 0 || a >= K + 2 || b >= K + 2) {
    dev_set_error(g.err, HAMSM_ERANGE, (int)i, a < 0 || a >= K + 2 ? a : b);
    return;
  }
  __hip_atomic_fetch_add(&part[xcc * cells + (size_t)a * (K + 2) + b], g.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ void k_flux_fold(double *__restrict__ F, const double *__restrict__ part, size_t cells) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cells) return;
  double s = part[c];
#pragma unroll
  for (int x = 1; x < FLUX_NXCD; x++) s = s + part[x * cells + c];
  if (s != 0.0) F[c] = F[c] + s;
}

// cells whose two sums differ by more than the reordering of ~10^4 f64 adds can explain
__global__ void k_flux_compare(const double *__restrict__ a, const double *__restrict__ b, size_t cells, unsigned long long *bad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  const double x = a[i], y = b[i], d = x > y ? x - y : y - x, m = (x < 0 ? -x : x) + (y < 0 ? -y : y);
  if (d > 1e-9 * m || x != x || y != y) atomicAdd(bad, 1ull);
}

int hamsm_flux_sorted(hamsm_ctx *c, const MapperDev &m, const FluxArgs &g, const int64_t *d_offs, int64_t n_iters);
int hamsm_flux_rows(hamsm_ctx *c, const MapperDev &m, const FluxArgs &g, const int64_t *d_offs, int64_t n_iters, int *taken);

extern "C" int hamsm_flux_accumulate(hamsm_ctx *c, const int32_t *pa, const int32_t *ch, const double *p0, const double *p1,
                                     const double *w, int64_t n, int K, double *F, int mode, const int64_t *d_offs,
                                     int64_t n_iters) {
  CTX(c);
  if (!c->have_bounds) return hamsm_fail(c, HAMSM_EINVAL, "bounds not set");
  if (n <= 0) return HAMSM_OK;
  int64_t i0 = 0;
  if (d_offs) {
    HIPCHK(c, hipMemcpyAsync(&i0, d_offs, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  FluxArgs g{pa, ch, p0, p1, w, i0, n, K, F, c->d_err};
  hamsm_timer_begin(c);
  int rc = HAMSM_OK;
  const size_t cells = (size_t)(K + 2) * (K + 2), part_bytes = sizeof(double) * cells * FLUX_NXCD;
  if (mode == 0 && c->flux_xcd == 1 && !c->flux_xcd_checked) {
    // The per-XCD path leans on two properties of this part that the HIP memory model does not promise: HW_REG_XCC_ID names
    // one of 8 XCDs, and atomics on ordinary hipMalloc memory are performed in the issuing XCD's L2 (so that partial
    // matrix x is only ever touched through L2 x and is whole at the kernel boundary).  It is therefore enabled only on the
    // device it was written for, and the first call that would use it is CHECKED against the single-matrix kernel.
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 || prop.multiProcessorCount != 256) c->flux_xcd = 0;
  }
  if (mode == 0 && c->flux_xcd && part_bytes <= FLUX_PART_MAX_BYTES && n >= 100000) {
    if (part_bytes > c->flux_part_bytes) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      hipFree(c->flux_part);
      c->flux_part = nullptr;
      c->flux_part_bytes = 0;
      hipError_t e = hipMalloc((void **)&c->flux_part, part_bytes);
      if (e != hipSuccess) return hamsm_fail(c, HAMSM_ENOMEM, "flux partials hipMalloc(%zu): %s", part_bytes, hipGetErrorString(e));
      c->flux_part_bytes = part_bytes;
    }
    double *Fchk = nullptr;
    if (!c->flux_xcd_checked) {  // reference result of this very call, by the device-scope kernel, on a copy of F
      rc = hamsm_ensure_scratch(c, sizeof(double) * (cells + 2));
      if (rc) return rc;
      Fchk = (double *)c->scratch;
      HIPCHK(c, hipMemcpyAsync(Fchk, F, sizeof(double) * cells, hipMemcpyDeviceToDevice, c->stream));
      FluxArgs g2 = g;
      g2.F = Fchk;
      hipLaunchKernelGGL(k_flux_atomic, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->mapper, g2);
    }
    HIPCHK(c, hipMemsetAsync(c->flux_part, 0, part_bytes, c->stream));
    hipLaunchKernelGGL(k_flux_xcd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->mapper, g, c->flux_part, cells);
    hipLaunchKernelGGL(k_flux_fold, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, c->stream, F, (const double *)c->flux_part, cells);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = hamsm_fail(c, HAMSM_EDEVICE, "k_flux_xcd: %s", hipGetErrorString(e));
    if (!rc && Fchk) {
      unsigned long long *bad = (unsigned long long *)(Fchk + cells);
      HIPCHK(c, hipMemsetAsync(bad, 0, sizeof(unsigned long long), c->stream));
      hipLaunchKernelGGL(k_flux_compare, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, c->stream, (const double *)F,
                         (const double *)Fchk, cells, bad);
      unsigned long long h_bad = 0;
      HIPCHK(c, hipMemcpyAsync(&h_bad, bad, sizeof h_bad, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      c->flux_xcd_checked = 1;
      if (h_bad) {  // the premise does not hold here: keep the device-scope result, never take this path again
        c->flux_xcd = 0;
        HIPCHK(c, hipMemcpyAsync(F, Fchk, sizeof(double) * cells, hipMemcpyDeviceToDevice, c->stream));
        fprintf(stderr, "[hamsm] per-XCD flux partials disagree with the device-scope kernel in %llu cells: path disabled\n", h_bad);
      }
    }
  } else if (mode == 0) {
    hipLaunchKernelGGL(k_flux_atomic, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->mapper, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = hamsm_fail(c, HAMSM_EDEVICE, "k_flux_atomic: %s", hipGetErrorString(e));
  } else {
    // deterministic: by rows (hand-written throughout, flux_rows.hip) where the matrix has few enough states, else by a stable
    // sort of the segments (flux_sorted.hip); the same bits either way
    int taken = 0;
    rc = hamsm_flux_rows(c, c->mapper, g, d_offs, n_iters, &taken);
    if (!rc && !taken) rc = hamsm_flux_sorted(c, c->mapper, g, d_offs, n_iters);
    c->flux_last_path = taken ? 2 : 1;
  }
  if (mode == 0) c->flux_last_path = 0;
  hamsm_timer_end(c, ST_FLUX);
  if (rc) return rc;
  return hamsm_check_device_error(c, "flux_accumulate");
}

// Not part of the product ABI (test hook): which deterministic path hamsm_flux_accumulate may take (rows: -1 leave, 0 off, 1 on;
// max_r: most states of the by-rows path, <= 0 leave); returns the path the LAST call took (0 atomics, 1 sorted, 2 by rows)
extern "C" int hamsm_debug_flux_path(hamsm_ctx *c, int rows, int max_r) {
  CTX(c);
  if (rows >= 0) c->flux_rows = rows != 0;
  if (max_r > 0) c->flux_rows_max_r = max_r;
  return c->flux_last_path;
}

__global__ void k_scale(double *F, int64_t n, double div) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) F[i] = F[i] / div;  // _fluxmatrix.py:342, true division
}

extern "C" int hamsm_scale(hamsm_ctx *c, double *F, int64_t count, double div) {
  CTX(c);
  if (count <= 0) return HAMSM_OK;
  hipLaunchKernelGGL(k_scale, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, F, count, div);
  HIPCHK(c, hipGetLastError());
  return HAMSM_OK;
}

// ---- a11 ----------------------------------------------------------------------------------------
// order-preserving map double <-> uint64 so that min/max are native integer atomics
__device__ __forceinline__ unsigned long long dkey(double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(unsigned long long k) {
  unsigned long long u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}

__global__ void k_stats_init(unsigned long long *kmin, unsigned long long *kmax, int ns) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ns) {
    kmin[i] = ~0ull;
    kmax[i] = 0ull;
  }
}

// Per-cluster pcoord sum / min / max / count.  Ten million frames fall on ~10^3 clusters, so global atomics collide
// ~10^4-fold per address (14 ms per pass); each workgroup therefore accumulates in LDS (ds_add_f64 / ds_min_u64 /
// ds_max_u64, 28 bytes per cluster) and flushes its touched entries once.  ns too large for LDS: global atomics.
#define STATS_LDS_MAX 2048  // 56 KB of LDS
#define STATS_PER_WG 16384
__global__ __launch_bounds__(256) void k_stats(const int32_t *__restrict__ lab, const double *__restrict__ p, int ndim, int64_t n,
                                               int ns, double *sum, unsigned long long *kmin, unsigned long long *kmax,
                                               unsigned long long *cnt, int *err, int use_lds) {
  extern __shared__ unsigned long long sh[];
  double *lsum = (double *)sh;
  unsigned long long *lmin = sh + ns, *lmax = sh + 2 * (size_t)ns;
  unsigned int *lcnt = (unsigned int *)(sh + 3 * (size_t)ns);
  if (use_lds) {
    for (int j = threadIdx.x; j < ns; j += 256) {
      lsum[j] = 0.0;
      lmin[j] = ~0ull;
      lmax[j] = 0ull;
      lcnt[j] = 0u;
    }
    __syncthreads();
  }
  const int64_t i0 = (int64_t)blockIdx.x * STATS_PER_WG, i1 = i0 + STATS_PER_WG < n ? i0 + STATS_PER_WG : n;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    int l = lab[i];
    if (l < 0 || l >= ns) {
      dev_set_error(err, HAMSM_ERANGE, (int)i, l);
      continue;
    }
    double v = p[i * ndim];
    if (use_lds) {
      unsafeAtomicAdd(&lsum[l], v);
      if (v == v) {  // NaN never wins a min/max (the oracle's < and > are false for NaN)
        atomicMin(&lmin[l], dkey(v));
        atomicMax(&lmax[l], dkey(v));
      }
      atomicAdd(&lcnt[l], 1u);
    } else {
      unsafeAtomicAdd(&sum[l], v);
      if (v == v) {
        atomicMin(&kmin[l], dkey(v));
        atomicMax(&kmax[l], dkey(v));
      }
      atomicAdd(&cnt[l], 1ull);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int j = threadIdx.x; j < ns; j += 256)
      if (lcnt[j]) {
        unsafeAtomicAdd(&sum[j], lsum[j]);
        if (lmin[j] != ~0ull) atomicMin(&kmin[j], lmin[j]);
        if (lmax[j] != 0ull) atomicMax(&kmax[j], lmax[j]);
        atomicAdd(&cnt[j], (unsigned long long)lcnt[j]);
      }
  }
}

__global__ void k_stats_fin(const unsigned long long *kmin, const unsigned long long *kmax, int ns, double *mn, double *mx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  if (kmin[i] != ~0ull) {
    double a = dunkey(kmin[i]);
    if (a < mn[i]) mn[i] = a;
  }
  if (kmax[i] != 0ull) {
    double b = dunkey(kmax[i]);
    if (b > mx[i]) mx[i] = b;
  }
}

extern "C" int hamsm_cluster_stats(hamsm_ctx *c, const int32_t *lab, const double *p, int64_t n, int ns, double *sum,
                                   double *mn, double *mx, int64_t *cnt) {
  CTX(c);
  if (n <= 0) return HAMSM_OK;
  int rc = hamsm_ensure_scratch(c, sizeof(unsigned long long) * 2 * (size_t)ns);
  if (rc) return rc;
  unsigned long long *kmin = (unsigned long long *)c->scratch, *kmax = kmin + ns;
  hamsm_timer_begin(c);
  hipLaunchKernelGGL(k_stats_init, dim3((ns + 255) / 256), dim3(256), 0, c->stream, kmin, kmax, ns);
  const int use_lds = ns <= STATS_LDS_MAX;
  const size_t lds = use_lds ? (size_t)ns * 28 + 8 : 0;
  hipLaunchKernelGGL(k_stats, dim3((unsigned)((n + STATS_PER_WG - 1) / STATS_PER_WG)), dim3(256), lds, c->stream, lab, p, c->ndim,
                     n, ns, sum, kmin, kmax, (unsigned long long *)cnt, c->d_err, use_lds);
  hipLaunchKernelGGL(k_stats_fin, dim3((ns + 255) / 256), dim3(256), 0, c->stream, kmin, kmax, ns, mn, mx);
  hamsm_timer_end(c, ST_STATS);
  HIPCHK(c, hipGetLastError());
  return hamsm_check_device_error(c, "cluster_stats");
}

// ---- a13 feed -----------------------------------------------------------------------------------
// numpy pairwise summation (what np.sum(fluxmatrix, 1) does per contiguous row), reproduced so that
// T is bit-equal to AnalysisMixin.get_Tmatrix.  One thread per row; depth <= log2(n/128)+1.
__device__ double np_pairwise(const double *a, int64_t n) {
  if (n < 8) {
    double r = -0.0;
    for (int64_t i = 0; i < n; i++) r = r + a[i];
    return r;
  } else if (n <= 128) {
    double r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; k++) r[k] = r[k] + a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + a[i];
    return res;
  } else {
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
  }
}

__device__ __forceinline__ double t_entry(const double *row, double out, int i, int j, bool is_t, const int32_t *basis,
                                          int nb) {
  if (is_t) {  // _analysis.py:74-77
    double v = 0.0;
    for (int k = 0; k < nb; k++)
      if (basis[k] == j) v = 1.0 / (double)nb;
    return v;
  }
  double v = row[j];
  if (out > 0) v = v / out;           // :53-56
  if (out == 0.0 && j == i) v = 1.0;  // :60-61
  return v;
}

__global__ void k_trow(const double *F, int ns, const int32_t *basis, int nb, const int32_t *target, int nt, double *rowsum,
                       int64_t *nnz) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const double *row = F + (size_t)i * ns;
  double out = np_pairwise(row, ns);
  rowsum[i] = out;
  bool is_t = false;
  for (int k = 0; k < nt; k++) is_t |= (target[k] == i);
  int64_t c = 0;
  for (int j = 0; j < ns; j++) c += t_entry(row, out, i, j, is_t, basis, nb) != 0.0;
  nnz[i] = c;
}

__global__ void k_tfill(const double *F, int ns, const int32_t *basis, int nb, const int32_t *target, int nt,
                        const double *rowsum, const int64_t *indptr, int32_t *indices, double *data) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const double *row = F + (size_t)i * ns;
  double out = rowsum[i];
  bool is_t = false;
  for (int k = 0; k < nt; k++) is_t |= (target[k] == i);
  int64_t o = indptr[i];
  for (int j = 0; j < ns; j++) {
    double v = t_entry(row, out, i, j, is_t, basis, nb);
    if (v != 0.0) {
      indices[o] = j;
      data[o] = v;
      o++;
    }
  }
}

extern "C" int hamsm_tmatrix_csr(hamsm_ctx *c, const double *F, int ns, const int32_t *basis, int nb, const int32_t *target,
                                 int nt, int64_t *indptr, int32_t *indices, double *data, int64_t cap, int64_t *nnz_out) {
  CTX(c);
  if (ns <= 0) return hamsm_fail(c, HAMSM_EINVAL, "nstates");
  size_t o_rs = 0, o_nz = o_rs + sizeof(double) * ns, o_ip = o_nz + sizeof(int64_t) * ns,
         o_bt = o_ip + sizeof(int64_t) * (ns + 1), o_end = o_bt + sizeof(int32_t) * (nb + nt + 2);
  int rc = hamsm_ensure_scratch(c, o_end);
  if (rc) return rc;
  char *base = (char *)c->scratch;
  double *rowsum = (double *)(base + o_rs);
  int64_t *d_nnz = (int64_t *)(base + o_nz), *d_ip = (int64_t *)(base + o_ip);
  int32_t *d_b = (int32_t *)(base + o_bt), *d_t = d_b + nb;
  HIPCHK(c, hipMemcpyAsync(d_b, basis, sizeof(int32_t) * nb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_t, target, sizeof(int32_t) * nt, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_trow, dim3((ns + 63) / 64), dim3(64), 0, c->stream, F, ns, d_b, nb, d_t, nt, rowsum, d_nnz);
  HIPCHK(c, hipGetLastError());
  std::vector<int64_t> cnt(ns);
  HIPCHK(c, hipMemcpyAsync(cnt.data(), d_nnz, sizeof(int64_t) * ns, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  indptr[0] = 0;
  for (int i = 0; i < ns; i++) indptr[i + 1] = indptr[i] + cnt[i];
  int64_t nnz = indptr[ns];
  *nnz_out = nnz;
  if (!indices && !data) return HAMSM_OK;  // count only
  if (nnz > cap) return hamsm_fail(c, HAMSM_ENOMEM, "csr capacity %lld too small for %lld", (long long)cap, (long long)nnz);
  HIPCHK(c, hipMemcpyAsync(d_ip, indptr, sizeof(int64_t) * (ns + 1), hipMemcpyHostToDevice, c->stream));
  int32_t *d_idx = nullptr;
  double *d_dat = nullptr;
  HIPCHK(c, hipMalloc((void **)&d_idx, sizeof(int32_t) * (size_t)(nnz ? nnz : 1)));
  HIPCHK(c, hipMalloc((void **)&d_dat, sizeof(double) * (size_t)(nnz ? nnz : 1)));
  hipLaunchKernelGGL(k_tfill, dim3((ns + 63) / 64), dim3(64), 0, c->stream, F, ns, d_b, nb, d_t, nt, rowsum, d_ip, d_idx,
                     d_dat);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(indices, d_idx, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(data, d_dat, sizeof(double) * nnz, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  hipFree(d_idx);
  hipFree(d_dat);
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "tmatrix_csr: %s", hipGetErrorString(e));
  return HAMSM_OK;
}

// ---- a12 on the device ------------------------------------------------------------------------------------------------------
__global__ void k_prow(const double *__restrict__ F, int ns, int64_t *nnz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const double *row = F + (size_t)i * ns;
  int64_t c = 0;
  for (int j = 0; j < ns; j++) c += row[j] != 0.0;
  nnz[i] = c;
}
__global__ void k_pfill(const double *__restrict__ F, int ns, const int64_t *__restrict__ indptr, int32_t *indices) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const double *row = F + (size_t)i * ns;
  int64_t o = indptr[i];
  for (int j = 0; j < ns; j++)
    if (row[j] != 0.0) indices[o++] = j;
}

extern "C" int hamsm_pattern_csr(hamsm_ctx *c, const double *F, int ns, int64_t *indptr, int32_t *indices, int64_t cap,
                                 int64_t *nnz_out) {
  CTX(c);
  if (ns <= 0 || !indptr || !nnz_out) return hamsm_fail(c, HAMSM_EINVAL, "pattern_csr: arguments");
  const size_t o_ip = (sizeof(int64_t) * ns + 255) & ~(size_t)255;
  int rc = hamsm_ensure_scratch(c, o_ip + sizeof(int64_t) * (ns + 1));
  if (rc) return rc;
  int64_t *d_nnz = (int64_t *)c->scratch, *d_ip = (int64_t *)((char *)c->scratch + o_ip);
  hipLaunchKernelGGL(k_prow, dim3((ns + 63) / 64), dim3(64), 0, c->stream, F, ns, d_nnz);
  HIPCHK(c, hipGetLastError());
  std::vector<int64_t> cnt(ns);
  hipError_t e = hipMemcpyAsync(cnt.data(), d_nnz, sizeof(int64_t) * ns, hipMemcpyDeviceToHost, c->stream);
  hipError_t es = hipStreamSynchronize(c->stream);
  if (e != hipSuccess || es != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "pattern_csr: read-back");
  indptr[0] = 0;
  for (int i = 0; i < ns; i++) indptr[i + 1] = indptr[i] + cnt[i];
  const int64_t nnz = indptr[ns];
  *nnz_out = nnz;
  if (!indices) return HAMSM_OK;
  if (nnz > cap) return hamsm_fail(c, HAMSM_ENOMEM, "csr capacity %lld too small for %lld", (long long)cap, (long long)nnz);
  if (nnz == 0) return HAMSM_OK;
  int32_t *d_idx = nullptr;
  HIPCHK(c, hipMemcpyAsync(d_ip, indptr, sizeof(int64_t) * (ns + 1), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMalloc((void **)&d_idx, sizeof(int32_t) * (size_t)nnz));
  hipLaunchKernelGGL(k_pfill, dim3((ns + 63) / 64), dim3(64), 0, c->stream, F, ns, d_ip, d_idx);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(indices, d_idx, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost, c->stream);
  es = hipStreamSynchronize(c->stream);
  hipFree(d_idx);
  if (e != hipSuccess || es != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "pattern_csr: %s", hipGetErrorString(e != hipSuccess ? e : es));
  return HAMSM_OK;
}

// np.sum over a whole array = numpy's reduction loop: the elements in MEMORY order, in chunks of the ufunc buffer size (8192
// elements, numpy's default `np.getbufsize()`), every chunk summed pairwise (np_pairwise: blocks of <= 128 with 8 partial sums,
// larger blocks halved at a multiple of 8) and the chunk sums added one after the other to a running total that starts at 0
// (checked against numpy 2.2 on C- and F-ordered matrices of 34^2 .. 1002^2 elements: identical bits).
#define NP_BUFSIZE 8192
__global__ void k_npsum_chunks(const double *__restrict__ a, long long n, double *__restrict__ part) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long off = k * NP_BUFSIZE;
  if (off >= n) return;
  part[k] = np_pairwise(a + off, n - off < NP_BUFSIZE ? n - off : NP_BUFSIZE);
}
__global__ void k_npsum_total(const double *__restrict__ part, long long nchunks, double *__restrict__ out) {
  if (blockIdx.x || threadIdx.x) return;
  double t = 0.0;
  for (long long k = 0; k < nchunks; k++) t = t + part[k];
  *out = t;
}

extern "C" int hamsm_np_sum(hamsm_ctx *c, const double *a, int64_t n, double *h_out) {
  CTX(c);
  if (!h_out) return hamsm_fail(c, HAMSM_EINVAL, "np_sum: out");
  if (n <= 0) {
    *h_out = 0.0;
    return HAMSM_OK;
  }
  const long long nch = (n + NP_BUFSIZE - 1) / NP_BUFSIZE;
  int rc = hamsm_ensure_scratch(c, sizeof(double) * (size_t)(nch + 1));
  if (rc) return rc;
  double *part = (double *)c->scratch;
  hipLaunchKernelGGL(k_npsum_chunks, dim3((unsigned)((nch + 63) / 64)), dim3(64), 0, c->stream, a, (long long)n, part);
  hipLaunchKernelGGL(k_npsum_total, dim3(1), dim3(64), 0, c->stream, part, nch, part + nch);
  HIPCHK(c, hipGetLastError());
  hipError_t e = hipMemcpyAsync(h_out, part + nch, sizeof(double), hipMemcpyDeviceToHost, c->stream);
  hipError_t es = hipStreamSynchronize(c->stream);
  if (e != hipSuccess || es != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "np_sum: read-back");
  return HAMSM_OK;
}

__global__ void k_permute_scale(const double *__restrict__ F, int ns, const int32_t *__restrict__ perm, double *__restrict__ out,
                                double div, int transpose) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= ns) return;
  const double v = transpose ? F[(size_t)perm[j] * ns + perm[i]] : F[(size_t)perm[i] * ns + perm[j]];
  out[(size_t)i * ns + j] = div == 1.0 ? v : v / div;  // true division (:1121)
}

extern "C" int hamsm_permute_scale(hamsm_ctx *c, const double *F, int ns, const int32_t *h_perm, double *out, double div,
                                   int transpose) {
  CTX(c);
  if (ns <= 0 || !h_perm || F == out) return hamsm_fail(c, HAMSM_EINVAL, "permute_scale: arguments");
  for (int i = 0; i < ns; i++)
    if (h_perm[i] < 0 || h_perm[i] >= ns) return hamsm_fail(c, HAMSM_EINVAL, "permute_scale: perm[%d] = %d", i, h_perm[i]);
  int rc = hamsm_ensure_scratch(c, sizeof(int32_t) * ns);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->scratch, h_perm, sizeof(int32_t) * ns, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_permute_scale, dim3((ns + 255) / 256, ns), dim3(256), 0, c->stream, F, ns, (const int32_t *)c->scratch, out, div, transpose);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // h_perm is the caller's
  return HAMSM_OK;
}

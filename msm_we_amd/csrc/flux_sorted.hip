// flux_sorted.hip -- deterministic flux accumulation (mode 1 of hamsm_flux_accumulate).
//
// Bit-identical to the reference's arithmetic: per iteration the COO->dense conversion adds duplicate
// (start, end) pairs in segment order, and the per-iteration matrices are added in ascending
// iteration order (_fluxmatrix.py:143-146, 325-327; oracle hamsm_flux_accumulate).  Realised as a
// stable radix sort of the segments by cell (rocPRIM, the one library primitive used) -- the segments
// arrive in ascending iteration, so a stable sort on the cell bits alone leaves every cell's segments in
// (iteration, segment) order; the key carries the iteration in its low bits, which the sort skips --
// then hand-written passes: per (cell, iteration) run an in-order partial sum, per cell an in-order sum
// of those partials on top of the value already in F.  Keys are 32-bit when cell and iteration fit.
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "common.h"

struct FluxArgs {
  const int32_t *pa, *ch;
  const double *p0, *p1, *w;
  int64_t i0, n;
  int K;
  double *F;
  int *err;
};

template <typename KT>
__global__ void k_flux_keys(MapperDev m, FluxArgs g, const int64_t *__restrict__ offs, int64_t n_iters, int ib,
                            KT *keys, double *vals) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // iteration of the workgroup's first segment by bisection, once; its other segments walk forward from there (an iteration
  // holds thousands of segments: at most a step or two) instead of 256 bisections of ~10 dependent loads each
  __shared__ long long s_it0;
  if (threadIdx.x == 0) {
    const int64_t i0 = g.i0 + (int64_t)blockIdx.x * blockDim.x;
    int64_t lo = 0, hi = n_iters;
    if (offs) {
      while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (offs[mid] <= i0) lo = mid;
        else hi = mid;
      }
    }
    s_it0 = lo;
  }
  __syncthreads();
  if (j >= g.n) return;
  int64_t i = g.i0 + j;
  double q0[HAMSM_MAX_NDIM], q1[HAMSM_MAX_NDIM];
  for (int d = 0; d < m.ndim; d++) {
    q0[d] = g.p0[i * m.ndim + d];
    q1[d] = g.p1[i * m.ndim + d];
  }
  int a = g.pa[i], b = g.ch[i];
  const int K = g.K;
  if (dev_in_box(m.target, m.ndim, q1)) b = K + 1;
  if (dev_in_box(m.basis, m.ndim, q0)) a = K;
  if (dev_in_box(m.basis, m.ndim, q1)) b = K;
  unsigned long long cell;
  if (a < 0 || b < 0 || a >= K + 2 || b >= K + 2) {
    dev_set_error(g.err, HAMSM_ERANGE, (int)i, a < 0 || a >= K + 2 ? a : b);
    cell = 0;
    vals[j] = 0.0;
  } else {
    cell = (unsigned long long)a * (K + 2) + b;
    vals[j] = g.w[i];
  }
  // iteration of segment i: last t with offs[t] <= i
  int64_t lo = s_it0;
  if (offs)
    while (lo + 1 < n_iters && offs[lo + 1] <= i) lo++;
  keys[j] = (KT)((cell << ib) | (unsigned long long)lo);
}

// pass 1: every (cell, iteration) run is summed in segment order (stable sort) by the thread at its first element, which also
// raises the run's flag -- and, in the upper half of the same word, the flag "first run of its cell"; pass 2 (after ONE exclusive
// scan of the 64-bit words: both running counts at once) packs the run sums, in order, into a dense array -- a cell's partials
// become CONTIGUOUS -- and notes where each cell's partials start; pass 3 is one thread per CELL adding its partials in ascending
// iteration on top of F.  (Walking the runs in place, `e += len[e]`, was a chain of up to n_iterations dependent loads per cell: 1.5
// ms of the 2.2 ms at C3.  Round 5: pass 3 used to be one thread per PARTIAL, of which the cell's first found the cell's end by
// galloping and bisecting -- ~14 dependent loads in one lane of 25, in every wave: 200 us at C3; with the cells' starts from the scan
// it is a dense launch over the cells.)
template <typename KT>
__global__ void k_flux_parts(const KT *__restrict__ ks, const double *__restrict__ ws, int64_t n, int ib,
                             double *part, unsigned long long *flag) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  // (everything a run of one element needs is requested at once: most runs are -- the walk below costs a dependent load per element)
  const KT k = ks[j];
  const KT kp = j > 0 ? ks[j - 1] : k;
  const KT kn = j + 1 < n ? ks[j + 1] : (KT)~k;
  const double w0 = ws[j];
  if (j > 0 && kp == k) {  // not the first element of its (cell, iteration) run
    flag[j] = 0;
    return;
  }
  double p = 0.0;
  p = p + w0;
  if (kn == k)
    for (int64_t e = j + 1; e < n && ks[e] == k; e++) p = p + ws[e];  // segment order inside the iteration (stable sort)
  part[j] = p;
  const bool cell_first = j == 0 || (kp >> ib) != (k >> ib);
  flag[j] = 1ull | ((unsigned long long)cell_first << 32);
}

template <typename KT>
__global__ void k_flux_pack(const KT *__restrict__ ks, const double *__restrict__ part,
                            const unsigned long long *__restrict__ flag, const unsigned long long *__restrict__ pos, int64_t n,
                            int ib, double *cpart, KT *ccell, int32_t *cfirst, int32_t *total) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long f = flag[j], p = pos[j];
  if (f) {
    cpart[(unsigned)p] = part[j];
    ccell[(unsigned)p] = ks[j] >> ib;
    if (f >> 32) cfirst[p >> 32] = (int32_t)(unsigned)p;
  }
  if (j == n - 1) {
    total[0] = (int32_t)((unsigned)p + (unsigned)f);   // partials
    total[1] = (int32_t)((p >> 32) + (f >> 32));       // cells
  }
}

template <typename KT>
__global__ void k_flux_cells(const KT *__restrict__ ccell, const double *__restrict__ cpart, const int32_t *__restrict__ cfirst,
                             const int32_t *__restrict__ total, double *F) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t M = total[0], NC = total[1];
  if (m >= NC) return;
  const int64_t lo = cfirst[m], hi = m + 1 < NC ? cfirst[m + 1] : M;  // the cell's partials: a KNOWN range, its loads do not wait
  const KT cell = ccell[lo];                                            // for each other
  double acc = F[cell];
  int64_t e = lo;
  for (; e + 32 <= hi; e += 32) {  // (a hot cell -- anything -> target: one partial per iteration -- is the launch's tail: 32 loads in flight)
    double v[32];
#pragma unroll
    for (int q = 0; q < 32; q++) v[q] = cpart[e + q];
#pragma unroll
    for (int q = 0; q < 32; q++) acc = acc + v[q];  // ascending iteration
  }
  for (; e + 8 <= hi; e += 8) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = cpart[e + q];
#pragma unroll
    for (int q = 0; q < 8; q++) acc = acc + v[q];  // ascending iteration
  }
  for (; e < hi; e++) acc = acc + cpart[e];
  F[cell] = acc;
}

template <typename KT>
static int flux_sorted_run(hamsm_ctx *c, const MapperDev &m, const FluxArgs &g, const int64_t *d_offs, int64_t n_iters, int ib, int cb) {
  const int64_t n = g.n;
  size_t temp_sort = 0, temp_scan = 0;
  KT *nullk = nullptr;
  double *nullv = nullptr;
  unsigned long long *nullu = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, temp_sort, nullk, nullk, nullv, nullv, (size_t)n, ib, ib + cb, c->stream);
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "radix_sort_pairs(size): %s", hipGetErrorString(e));
  e = rocprim::exclusive_scan(nullptr, temp_scan, nullu, nullu, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), c->stream);
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "exclusive_scan(size): %s", hipGetErrorString(e));
  size_t temp_bytes = temp_sort > temp_scan ? temp_sort : temp_scan;
  size_t a8 = ((size_t)n * 8 + 255) & ~(size_t)255, a4 = ((size_t)n * 4 + 255) & ~(size_t)255;
  // k_in | k_out | v_in (later: cpart) | v_out | part | flag | pos | cfirst | total | temp ; ccell reuses k_in
  size_t need = 7 * a8 + a4 + 256 + temp_bytes + 256;
  int rc = hamsm_ensure_scratch(c, need);
  if (rc) return rc;
  char *base = (char *)c->scratch;
  KT *k_in = (KT *)base, *k_out = (KT *)(base + a8);
  double *v_in = (double *)(base + 2 * a8), *v_out = (double *)(base + 3 * a8), *part = (double *)(base + 4 * a8);
  unsigned long long *flag = (unsigned long long *)(base + 5 * a8), *pos = (unsigned long long *)(base + 6 * a8);
  int32_t *cfirst = (int32_t *)(base + 7 * a8), *total = (int32_t *)(base + 7 * a8 + a4);
  void *temp = base + 7 * a8 + a4 + 256;
  unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_flux_keys<KT>, dim3(grid), dim3(256), 0, c->stream, m, g, d_offs, n_iters, ib, k_in, v_in);
  // (stable, and only over the cell bits: the iteration bits below them are already in ascending order)
  e = rocprim::radix_sort_pairs(temp, temp_bytes, k_in, k_out, v_in, v_out, (size_t)n, ib, ib + cb, c->stream);
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "radix_sort_pairs: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(k_flux_parts<KT>, dim3(grid), dim3(256), 0, c->stream, k_out, v_out, n, ib, part, flag);
  e = rocprim::exclusive_scan(temp, temp_bytes, flag, pos, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), c->stream);
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "exclusive_scan: %s", hipGetErrorString(e));
  double *cpart = v_in;  // (the unsorted values are dead)
  KT *ccell = k_in;      // (so are the unsorted keys)
  hipLaunchKernelGGL(k_flux_pack<KT>, dim3(grid), dim3(256), 0, c->stream, k_out, part, flag, pos, n, ib, cpart, ccell, cfirst, total);
  // one thread per cell: at most (K + 2)^2 cells, at most n
  const unsigned long long cells_max = (unsigned long long)(g.K + 2) * (g.K + 2);
  const unsigned gridc = (unsigned)(((cells_max < (unsigned long long)n ? cells_max : (unsigned long long)n) + 255) / 256);
  hipLaunchKernelGGL(k_flux_cells<KT>, dim3(gridc), dim3(256), 0, c->stream, ccell, cpart, cfirst, total, g.F);
  e = hipGetLastError();
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "flux_sorted: %s", hipGetErrorString(e));
  return HAMSM_OK;
}

int hamsm_flux_sorted(hamsm_ctx *c, const MapperDev &m, const FluxArgs &g, const int64_t *d_offs, int64_t n_iters) {
  if (!d_offs) n_iters = 1;
  if (g.n > 0x7fffffffLL) return hamsm_fail(c, HAMSM_ENOTSUP, "deterministic flux: more than 2^31 segments per call");
  // key = cell << ib | iteration: ib bits hold 0 .. n_iters - 1, cb bits hold 0 .. (K + 2)^2 - 1
  const unsigned long long cells = (unsigned long long)(g.K + 2) * (g.K + 2);
  int ib = 0, cb = 1;
  while (ib < 62 && ((unsigned long long)(n_iters - 1) >> ib)) ib++;
  while (cb < 62 && ((cells - 1) >> cb)) cb++;
  if (ib + cb > 64) return hamsm_fail(c, HAMSM_ENOTSUP, "deterministic flux: %d cell bits + %d iteration bits", cb, ib);
  if (ib + cb <= 32) return flux_sorted_run<unsigned>(c, m, g, d_offs, n_iters, ib, cb);
  return flux_sorted_run<unsigned long long>(c, m, g, d_offs, n_iters, ib, cb);
}

// ---- f4: stable grouping of segment indices by label -----------------------------------------------------------------------
__global__ void k_group_keys(const int32_t *__restrict__ lab, int64_t n, int ns, unsigned *keys, int32_t *vals,
                             unsigned long long *cnt, int *err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int l = lab[i];
  if (l < 0 || l >= ns) {
    dev_set_error(err, HAMSM_ERANGE, (int)i, l);
    l = 0;
  } else {
    atomicAdd(&cnt[l], 1ull);
  }
  keys[i] = (unsigned)l;
  vals[i] = (int32_t)i;
}

extern "C" int hamsm_group_by_label(hamsm_ctx *c, const int32_t *lab, int64_t n, int ns, int32_t *order, int64_t *h_indptr) {
  CTX(c);
  if (ns <= 0 || !h_indptr) return hamsm_fail(c, HAMSM_EINVAL, "group_by_label: nstates / indptr");
  for (int s = 0; s <= ns; s++) h_indptr[s] = 0;
  if (n <= 0) return HAMSM_OK;
  if (n > 0x7fffffffLL) return hamsm_fail(c, HAMSM_ENOTSUP, "group_by_label: more than 2^31 segments");
  int bits = 1;
  while (bits < 32 && ((unsigned)ns >> bits)) bits++;
  size_t temp_bytes = 0;
  unsigned *nk = nullptr;
  int32_t *nv = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, temp_bytes, nk, nk, nv, nv, (size_t)n, 0, bits, c->stream);
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "radix_sort_pairs(size): %s", hipGetErrorString(e));
  const size_t a4 = ((size_t)n * 4 + 255) & ~(size_t)255, ac = ((size_t)ns * 8 + 255) & ~(size_t)255;
  int rc = hamsm_ensure_scratch(c, 3 * a4 + ac + temp_bytes + 256);
  if (rc) return rc;
  char *base = (char *)c->scratch;
  unsigned *k_in = (unsigned *)base, *k_out = (unsigned *)(base + a4);
  int32_t *v_in = (int32_t *)(base + 2 * a4);
  unsigned long long *cnt = (unsigned long long *)(base + 3 * a4);
  void *temp = base + 3 * a4 + ac;
  HIPCHK(c, hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * ns, c->stream));
  hipLaunchKernelGGL(k_group_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, lab, n, ns, k_in, v_in, cnt, c->d_err);
  e = rocprim::radix_sort_pairs(temp, temp_bytes, k_in, k_out, v_in, order, (size_t)n, 0, bits, c->stream);  // stable
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "radix_sort_pairs: %s", hipGetErrorString(e));
  std::vector<unsigned long long> hc((size_t)ns);
  e = hipMemcpyAsync(hc.data(), cnt, sizeof(unsigned long long) * ns, hipMemcpyDeviceToHost, c->stream);
  hipError_t es = hipStreamSynchronize(c->stream);
  if (e != hipSuccess || es != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "group_by_label read-back");
  for (int s = 0; s < ns; s++) h_indptr[s + 1] = h_indptr[s] + (int64_t)hc[s];
  return hamsm_check_device_error(c, "group_by_label");
}

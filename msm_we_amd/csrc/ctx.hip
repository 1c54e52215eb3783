// ctx.hip -- lifecycle, device memory, model state (mapper, bounds, remap, centroids).
#include <stdlib.h>

#include <mutex>
#include <unordered_set>

#include "common.h"

// ---- handle registry ------------------------------------------------------------------------------
// hamsm_destroy releases everything a context owns but never returns the struct itself to the heap: a handle stays
// recognisable (HAMSM_CTX_DEAD) for the life of the process, so a stale wrapper -- a second destroy, a buffer freed
// after its context -- gets HAMSM_EINVAL instead of acting on whatever the allocator placed at that address next.
static std::mutex g_reg_mu;
static std::unordered_set<const hamsm_ctx *> &registry() {
  static std::unordered_set<const hamsm_ctx *> *r = new std::unordered_set<const hamsm_ctx *>();  // never destructed
  return *r;
}
static thread_local char g_stale_msg[160];

static bool ctx_known(const hamsm_ctx *c) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  return registry().count(c) != 0;
}

int hamsm_ctx_check(hamsm_ctx *c, const char *fn) {
  if (!c) return HAMSM_EINVAL;
  if (!ctx_known(c)) {
    snprintf(g_stale_msg, sizeof g_stale_msg, "%s: %p is not a hamsm context of this library", fn, (void *)c);
    return HAMSM_EINVAL;
  }
  if (c->magic == HAMSM_CTX_DEAD) {
    snprintf(g_stale_msg, sizeof g_stale_msg, "%s: stale handle %p (context was destroyed)", fn, (void *)c);
    return HAMSM_EINVAL;
  }
  if (c->magic != HAMSM_CTX_LIVE) {
    snprintf(g_stale_msg, sizeof g_stale_msg, "%s: context %p OVERWRITTEN (magic %llx)", fn, (void *)c, (unsigned long long)c->magic);
    return HAMSM_EINVAL;
  }
  return HAMSM_OK;
}

int hamsm_fail(hamsm_ctx *c, int code, const char *fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
  }
  return code;
}

extern "C" int hamsm_abi_version(void) { return HAMSM_ABI_VERSION; }
extern "C" const char *hamsm_backend_name(void) { return "hip-gfx950"; }

extern "C" int hamsm_create(int device, hamsm_ctx **out) {
  if (!out) return HAMSM_EINVAL;
  hamsm_ctx *c = new hamsm_ctx();
  c->err[0] = 0;
  c->device = device;
  memset(&c->mapper, 0, sizeof c->mapper);
  if (const char *v = getenv("HAMSM_ASSIGN_W")) c->assign_waves = atoi(v) == 8 ? 8 : 4;
  if (const char *v = getenv("HAMSM_SEED_G_MB")) c->seed_g_budget = (size_t)(atoi(v) < 1 ? 1 : atoi(v)) << 20;
  if (const char *v = getenv("HAMSM_FILTER_BF16")) c->filter_bf16 = atoi(v) < 0 ? 0 : (atoi(v) > 2 ? 2 : atoi(v));
  if (const char *v = getenv("HAMSM_FLUX_XCD")) c->flux_xcd = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_ASSIGN_FILTER")) c->assign_filter = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_FILTER_LITE")) c->filter_lite = atoi(v) < 0 ? 0 : (atoi(v) > 2 ? 2 : atoi(v));
  if (const char *v = getenv("HAMSM_FILTER_DBUF")) c->filter_dbuf = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_FILTER_NB")) c->filter_nb = atoi(v) < 1 ? 1 : (atoi(v) > 16 ? 16 : atoi(v));
  if (const char *v = getenv("HAMSM_FILTER_LDS_KB")) c->filter_lds_kb = atoi(v) < 40 ? 40 : (atoi(v) > 150 ? 150 : atoi(v));
  if (const char *v = getenv("HAMSM_GENERIC_STAGE_KB")) c->generic_stage_kb = atoi(v) < 0 ? 0 : (atoi(v) > 150 ? 150 : atoi(v));
  if (const char *v = getenv("HAMSM_STEP_RANKED")) c->step_ranked = atoi(v) < 0 ? 0 : (atoi(v) > 2 ? 2 : atoi(v));
  if (const char *v = getenv("HAMSM_STEP_DIRECT")) c->step_direct = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_EXACT_CAPPED")) c->exact_capped = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_STEP_DIRECT_PLAIN")) c->step_direct_plain = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_STEP_LROWS")) c->step_lrows = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_ENQUEUE_THREADS")) c->enqueue_threads = atoi(v) < 1 ? 1 : atoi(v);
  if (const char *v = getenv("HAMSM_FLUX_ROWS")) c->flux_rows = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_FLUX_ROWS_MAX_R")) c->flux_rows_max_r = atoi(v) < 2 ? 2 : atoi(v);
  if (const char *v = getenv("HAMSM_STEP_PIPE")) c->step_pipe = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_STEP_WIDE")) c->step_wide = atoi(v) < 0 ? 0 : (atoi(v) > 3 ? 3 : atoi(v));
  if (const char *v = getenv("HAMSM_STEP_RANKED_MFLOP")) c->step_ranked_min_flop = atof(v) > 0 ? atof(v) * 1e6 : c->step_ranked_min_flop;
#ifdef HAMSM_EXPERIMENTS  // (libhamsm_hip_exp.so only: csrc/experiments/)
  if (const char *v = getenv("HAMSM_TRAIN_PERSIST")) c->train_persist = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_PERSIST_WPX")) c->persist_wpx = atoi(v) < 4 ? 4 : (atoi(v) > 128 ? 128 : atoi(v));
  if (const char *v = getenv("HAMSM_TRAIN_RANKED")) c->train_ranked = atoi(v) < 0 ? 0 : (atoi(v) > 2 ? 2 : atoi(v));
  if (const char *v = getenv("HAMSM_RANKED_MIN_MFLOP")) c->ranked_min_flop = atof(v) > 0 ? atof(v) * 1e6 : c->ranked_min_flop;
  if (const char *v = getenv("HAMSM_TRAIN_GRAPH")) c->train_graph = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_TRAIN_AHEAD")) c->ahead = atoi(v) != 0;
  if (const char *v = getenv("HAMSM_AHEAD_RT")) c->ahead_rt = atoi(v) == 2 ? 2 : 1;
  if (const char *v = getenv("HAMSM_AHEAD_N0")) c->ahead_n0 = atoi(v) < 1 ? 1 : atoi(v);
  if (const char *v = getenv("HAMSM_AHEAD_G")) c->ahead_gmax = atoi(v) < 1 ? 1 : (atoi(v) > 16 ? 16 : atoi(v));
  if (const char *v = getenv("HAMSM_AHEAD_ALPHA")) c->ahead_alpha = atof(v) > 1.0 ? atof(v) : 1.0;
#endif
  if (const char *v = getenv("HAMSM_STEP_LANES")) c->step_lanes = atoi(v) < 1 ? 1 : (atoi(v) > 4 ? 4 : atoi(v));
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void **)&c->d_err, 4 * sizeof(int)) != hipSuccess ||
      hipHostMalloc((void **)&c->h_err, 4 * sizeof(int)) != hipSuccess) {
    // no GPU, no product: the caller gets an error, never a CPU path
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->d_err) hipFree(c->d_err);
    delete c;  // (never registered: no handle was handed out)
    return HAMSM_EDEVICE;
  }
  hipMemsetAsync(c->d_err, 0, 4 * sizeof(int), c->stream);
  hipStreamSynchronize(c->stream);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->n_cu = cus;
  }
  c->magic = HAMSM_CTX_LIVE;
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    registry().insert(c);
  }
  *out = c;
  return HAMSM_OK;
}

static void free_strata(hamsm_ctx *c) {
  for (auto &t : c->st) {
    hipFree(t.C);
    hipFree(t.csq);
    hipFree(t.counts);
    hipFree(t.Bpack);
    t.C = t.csq = t.counts = t.Bpack = nullptr;
  }
  c->st.clear();
  hipFree(c->d_table);
  c->d_table = nullptr;
  hipFree(c->d_remap);
  c->d_remap = nullptr;
}

extern "C" int hamsm_comm_destroy(hamsm_ctx *c);
static void blk_cache_flush(hamsm_ctx *c);

template <typename T>
static void release(T &container) {
  T().swap(container);
}

extern "C" void hamsm_destroy(hamsm_ctx *c) {
  if (!c || !ctx_known(c) || c->magic != HAMSM_CTX_LIVE) return;  // foreign or already destroyed: nothing is freed twice
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  for (int i = 0; i < 3; i++)
    if (c->side_stream[i]) hipStreamSynchronize(c->side_stream[i]);
  hamsm_comm_destroy(c);
  blk_cache_flush(c);
  if (c->copy_stream) {
    hipStreamSynchronize(c->copy_stream);
    hipStreamDestroy(c->copy_stream);
    c->copy_stream = nullptr;
  }
  for (auto &kv : c->host_live) hipHostFree(kv.first);
  release(c->host_live);
  for (auto &kv : c->blk_live) hipFree(kv.first);  // device memory dies with its context
  free_strata(c);
  hamsm_pca_release(c);
  hipFree(c->d_bounds);
  hipFree(c->d_centers);
  hipFree(c->d_ext_bins_own);
  hipFree(c->d_err);
  hipHostFree(c->h_err);
  hipHostFree(c->h_und);
  hipFree(c->scratch);
  hipFree(c->scratch2);
  hipFree(c->flux_part);
  hipHostFree(c->h_pinned);
  for (auto &p : c->ev_pending) {
    hipEventDestroy(p.e0);
    hipEventDestroy(p.e1);
  }
  for (auto e : c->ev_free) hipEventDestroy(e);
  for (auto &k : c->ks_pending) {
    hipEventDestroy(k.e0);
    hipEventDestroy(k.e1);
    hipEventDestroy(k.e2);
  }
  for (auto e : c->ks_free) hipEventDestroy(e);
  release(c->ks_free);
  release(c->ks_pending);
  hipFree(c->ks_stamps);
  c->ks_stamps = nullptr;
  for (int i = 0; i < 3; i++) {
    if (c->side_stream[i]) hipStreamDestroy(c->side_stream[i]);
    if (c->ev_join[i]) hipEventDestroy(c->ev_join[i]);
  }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  for (int l = 0; l < 4; l++)
    for (int k = 0; k < 4; k++) {
      if (c->ev_ring[l][k]) hipEventDestroy(c->ev_ring[l][k]);
      c->ev_ring[l][k] = nullptr;
    }
  if (c->poll_stream) hipStreamDestroy(c->poll_stream);
  c->poll_stream = nullptr;
  if (c->h_poll) hipHostFree(c->h_poll);
  c->h_poll = nullptr;
  c->h_poll_n = 0;
  hipStreamDestroy(c->stream);
  // the struct stays (see the registry note): drop what it owns on the heap, null every pointer, mark it dead
  release(c->blk_live);
  release(c->blk_cache);
  release(c->h_bounds);
  release(c->st);
  release(c->ev_pending);
  release(c->ev_free);
  c->d_bounds = nullptr;
  c->d_centers = nullptr;
  c->d_ext_bins_own = nullptr;
  c->d_err = c->h_err = nullptr;
  c->h_und = nullptr;
  c->scratch = c->scratch2 = nullptr;
  c->scratch_bytes = c->scratch2_bytes = 0;
  c->flux_part = nullptr;
  c->h_pinned = nullptr;
  c->stream = nullptr;
  for (int i = 0; i < 3; i++) {
    c->side_stream[i] = nullptr;
    c->ev_join[i] = nullptr;
  }
  c->ev_fork = nullptr;
  c->have_mapper = c->have_bounds = false;
  c->magic = HAMSM_CTX_DEAD;
}

extern "C" const char *hamsm_last_error(hamsm_ctx *c) {
  if (!c) return "null ctx";
  if (!ctx_known(c) || c->magic != HAMSM_CTX_LIVE) return g_stale_msg[0] ? g_stale_msg : "stale or foreign ctx handle";
  return c->err;
}

// ---- memory -------------------------------------------------------------------------------------
static const size_t BLK_CACHE_MAX = (size_t)8 << 30;  // bytes kept for reuse (of 288 GB)

static void blk_cache_flush(hamsm_ctx *c) {
  if (c->blk_cache.empty()) return;
  hipStreamSynchronize(c->stream);
  for (auto &kv : c->blk_cache) hipFree(kv.second);
  c->blk_cache.clear();
  c->blk_cached_bytes = 0;
}

extern "C" int hamsm_malloc(hamsm_ctx *c, size_t bytes, void **out) {
  CTX(c);
  HIPCHK(c, hipSetDevice(c->device));
  const size_t want = ((bytes ? bytes : 1) + 255) & ~(size_t)255;
  auto it = c->blk_cache.lower_bound(want);
  if (it != c->blk_cache.end() && it->first <= want + want / 2 + (1 << 20)) {  // close enough in size: reuse
    *out = it->second;
    c->blk_live[it->second] = it->first;
    c->blk_cached_bytes -= it->first;
    c->blk_cache.erase(it);
    return HAMSM_OK;
  }
  hipError_t e = hipMalloc(out, want);
  if (e != hipSuccess) {
    blk_cache_flush(c);
    e = hipMalloc(out, want);
  }
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  c->blk_live[*out] = want;
  return HAMSM_OK;
}
extern "C" int hamsm_free(hamsm_ctx *c, void *p) {
  CTX(c);
  if (!p) return HAMSM_OK;
  auto it = c->blk_live.find(p);
  if (it == c->blk_live.end()) return hamsm_fail(c, HAMSM_EINVAL, "hamsm_free: %p was not returned by hamsm_malloc", p);
  const size_t sz = it->second;
  c->blk_live.erase(it);
  if (sz <= BLK_CACHE_MAX / 4 && c->blk_cached_bytes + sz <= BLK_CACHE_MAX) {
    c->blk_cache.emplace(sz, p);
    c->blk_cached_bytes += sz;
    return HAMSM_OK;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipFree(p));
  return HAMSM_OK;
}
extern "C" int hamsm_memcpy_h2d(hamsm_ctx *c, void *d, const void *h, size_t n) {
  CTX(c);
  HIPCHK(c, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the host buffer is the caller's: it may be reused on return
  return HAMSM_OK;
}
extern "C" int hamsm_memcpy_d2h(hamsm_ctx *c, void *h, const void *d, size_t n) {
  CTX(c);
  HIPCHK(c, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HAMSM_OK;
}
extern "C" int hamsm_malloc_host(hamsm_ctx *c, size_t bytes, void **out) {
  CTX(c);
  if (!out) return hamsm_fail(c, HAMSM_EINVAL, "malloc_host: out");
  HIPCHK(c, hipSetDevice(c->device));
  void *p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_ENOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
  c->host_live[p] = bytes;
  *out = p;
  return HAMSM_OK;
}
extern "C" int hamsm_free_host(hamsm_ctx *c, void *p) {
  CTX(c);
  if (!p) return HAMSM_OK;
  auto it = c->host_live.find(p);
  if (it == c->host_live.end()) return hamsm_fail(c, HAMSM_EINVAL, "hamsm_free_host: %p was not returned by hamsm_malloc_host", p);
  if (c->copy_stream) HIPCHK(c, hipStreamSynchronize(c->copy_stream));  // no upload may still read from it
  c->host_live.erase(it);
  HIPCHK(c, hipHostFree(p));
  return HAMSM_OK;
}
// (ABI 9) process-level page-locked blocks: owned by the caller's arrays, not by a context (hamsm.h)
extern "C" int hamsm_host_block_alloc(size_t bytes, void **out) {
  if (!out) return HAMSM_EINVAL;
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    return HAMSM_ENOMEM;
  }
  *out = p;
  return HAMSM_OK;
}
extern "C" int hamsm_host_block_free(void *p) {
  if (!p) return HAMSM_OK;
  return hipHostFree(p) == hipSuccess ? HAMSM_OK : HAMSM_EINVAL;
}
extern "C" int hamsm_memcpy_h2d_async(hamsm_ctx *c, void *d, const void *h, size_t n) {
  CTX(c);
  if (n == 0) return HAMSM_OK;
  bool pinned = false;  // a pageable source would make the call synchronous and its lifetime the caller's problem: refuse
  for (auto &kv : c->host_live)
    if ((const char *)h >= (const char *)kv.first && (const char *)h + n <= (const char *)kv.first + kv.second) pinned = true;
  if (!pinned) return hamsm_fail(c, HAMSM_EINVAL, "memcpy_h2d_async: the source is not inside a hamsm_malloc_host block");
  if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  HIPCHK(c, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->copy_stream));
  return HAMSM_OK;
}
extern "C" int hamsm_copy_sync(hamsm_ctx *c) {
  CTX(c);
  if (c->copy_stream) HIPCHK(c, hipStreamSynchronize(c->copy_stream));
  return HAMSM_OK;
}
// HBM-bound stream: 4 B in, 8 B out per element; four elements per thread (one 16-byte load, two 16-byte stores) while
// both pointers are 16-byte aligned, grid-stride so that any count fits one launch
__global__ void k_widen_f32(const float *__restrict__ src, double *__restrict__ dst, int64_t n) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x, t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t n4 = (((uintptr_t)src | (uintptr_t)dst) & 15) ? 0 : n / 4;
  for (int64_t i = t; i < n4; i += stride) {
    float4 v = reinterpret_cast<const float4 *>(src)[i];
    reinterpret_cast<double2 *>(dst)[2 * i] = make_double2((double)v.x, (double)v.y);
    reinterpret_cast<double2 *>(dst)[2 * i + 1] = make_double2((double)v.z, (double)v.w);
  }
  for (int64_t i = 4 * n4 + t; i < n; i += stride) dst[i] = (double)src[i];
}
extern "C" int hamsm_widen_f32(hamsm_ctx *c, const float *d_src, double *d_dst, int64_t count) {
  CTX(c);
  if (count < 0 || (count > 0 && (!d_src || !d_dst))) return hamsm_fail(c, HAMSM_EINVAL, "widen_f32: null buffer or negative count");
  if (count == 0) return HAMSM_OK;
  const char *s0 = (const char *)d_src, *d0 = (const char *)d_dst;
  if (s0 < d0 + 8 * count && d0 < s0 + 4 * count) return hamsm_fail(c, HAMSM_EINVAL, "widen_f32: source and destination overlap");
  int64_t blocks = (count / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 64) blocks = 256 * 64;  // 64 workgroups per CU, then the grid strides
  hipLaunchKernelGGL(k_widen_f32, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_src, d_dst, count);
  HIPCHK(c, hipGetLastError());
  return HAMSM_OK;
}
// rows that hold a NaN: one wave per row at a time (lane l reads elements l, l + 64, ...; a ballot decides), grid-stride over the
// rows.  HBM-bound (every byte read once, one byte per row written).
template <typename XT>
__global__ __launch_bounds__(256) void k_nan_rows(const XT *__restrict__ X, int64_t ldx, int D, int64_t n, uint8_t *__restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < n; i += nwaves) {
    const XT *row = X + i * ldx;
    bool bad = false;
    for (int k = lane; k < D; k += 64) {
      const XT v = row[k];
      bad = bad || (v != v);
    }
    const unsigned long long any = __ballot(bad);
    if (lane == 0) flags[i] = any ? 1 : 0;
  }
}
extern "C" int hamsm_nan_rows(hamsm_ctx *c, const void *d_X, int dt, int64_t ldx, int D, int64_t n, uint8_t *d_flags) {
  CTX(c);
  if (dt != HAMSM_F32 && dt != HAMSM_F64) return hamsm_fail(c, HAMSM_EINVAL, "x_dtype");
  if (n < 0 || D < 1 || ldx < D || (n > 0 && (!d_X || !d_flags))) return hamsm_fail(c, HAMSM_EINVAL, "nan_rows: bad arguments");
  if (n == 0) return HAMSM_OK;
  int64_t blocks = (n + 3) / 4;  // 4 waves per workgroup, a row per wave and turn
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (dt == HAMSM_F32) hipLaunchKernelGGL(k_nan_rows<float>, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const float *)d_X, ldx, D, n, d_flags);
  else hipLaunchKernelGGL(k_nan_rows<double>, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const double *)d_X, ldx, D, n, d_flags);
  HIPCHK(c, hipGetLastError());
  return HAMSM_OK;
}
extern "C" int hamsm_memset(hamsm_ctx *c, void *d, int b, size_t n) {
  CTX(c);
  HIPCHK(c, hipMemsetAsync(d, b, n, c->stream));
  return HAMSM_OK;
}
extern "C" int hamsm_sync(hamsm_ctx *c) {
  CTX(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return hamsm_check_device_error(c, "sync");
}

int hamsm_ensure_scratch(hamsm_ctx *c, size_t bytes) {
  c->scratch_epoch++;  // whoever asks for the arena is about to write into it: what an earlier call left there is gone
  if (bytes <= c->scratch_bytes) return HAMSM_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(c->scratch);
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  size_t want = bytes + bytes / 4 + 4096;
  hipError_t e = hipMalloc(&c->scratch, want);
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_ENOMEM, "scratch hipMalloc(%zu): %s", want, hipGetErrorString(e));
  c->scratch_bytes = want;
  return HAMSM_OK;
}

int hamsm_ensure_scratch2(hamsm_ctx *c, size_t bytes) {
  if (bytes <= c->scratch2_bytes) return HAMSM_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(c->scratch2);
  c->scratch2 = nullptr;
  c->scratch2_bytes = 0;
  size_t want = bytes + bytes / 4 + 4096;
  hipError_t e = hipMalloc(&c->scratch2, want);
  if (e != hipSuccess) return hamsm_fail(c, HAMSM_ENOMEM, "scratch2 hipMalloc(%zu): %s", want, hipGetErrorString(e));
  c->scratch2_bytes = want;
  return HAMSM_OK;
}

int hamsm_ensure_pinned(hamsm_ctx *c, size_t bytes) {
  if (bytes <= c->h_pinned_bytes) return HAMSM_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipHostFree(c->h_pinned);
  c->h_pinned = nullptr;
  c->h_pinned_bytes = 0;
  size_t want = bytes * 2 + 4096;
  HIPCHK(c, hipHostMalloc(&c->h_pinned, want));
  c->h_pinned_bytes = want;
  return HAMSM_OK;
}

int hamsm_check_device_error(hamsm_ctx *c, const char *what) {
  HIPCHK(c, hipMemcpyAsync(c->h_err, c->d_err, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int code = c->h_err[0];
  if (code == 0) return HAMSM_OK;
  int a = c->h_err[1], b = c->h_err[2];
  hipMemsetAsync(c->d_err, 0, 4 * sizeof(int), c->stream);
  switch (code) {
    case HAMSM_EUNFIT: return hamsm_fail(c, code, "%s: Not initialized in seg %d, bin %d", what, a, b);
    case HAMSM_ERANGE: return hamsm_fail(c, code, "%s: index out of range at element %d (value %d)", what, a, b);
    default: return hamsm_fail(c, code, "%s: device error %d (%d, %d)", what, code, a, b);
  }
}

// Timers: event pairs are recorded on the stream and resolved lazily, so timing never adds a host
// sync to the launch path (only hamsm_timers / hamsm_last_kernel_ms block).
static void timers_resolve(hamsm_ctx *c, bool block) {
  size_t done = 0;
  for (; done < c->ev_pending.size(); done++) {
    auto &p = c->ev_pending[done];
    if (block) hipEventSynchronize(p.e1);
    else if (hipEventQuery(p.e1) != hipSuccess) break;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      c->timers[p.stage] += ms * 1e-3;
      c->last_ms[p.stage] = ms;
    }
    c->ev_free.push_back(p.e0);
    c->ev_free.push_back(p.e1);
  }
  c->ev_pending.erase(c->ev_pending.begin(), c->ev_pending.begin() + done);
}

static hipEvent_t ev_get(hamsm_ctx *c) {
  if (!c->ev_free.empty()) {
    hipEvent_t e = c->ev_free.back();
    c->ev_free.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  hipEventCreate(&e);
  return e;
}

void hamsm_timer_begin(hamsm_ctx *c) {
  c->ev_cur0 = ev_get(c);
  hipEventRecord(c->ev_cur0, c->stream);
}
void hamsm_timer_end(hamsm_ctx *c, int stage) {
  hipEvent_t e1 = ev_get(c);
  hipEventRecord(e1, c->stream);
  c->ev_pending.push_back({c->ev_cur0, e1, stage});
  if (c->ev_pending.size() > 512) timers_resolve(c, false);
  if (c->ev_pending.size() > 8192) timers_resolve(c, true);
}

extern "C" int hamsm_timers(hamsm_ctx *c, double *out, int reset) {
  CTX(c);
  timers_resolve(c, true);
  for (int i = 0; i < ST_N; i++) out[i] = c->timers[i];
  if (reset)
    for (int i = 0; i < ST_N; i++) c->timers[i] = 0;
  return HAMSM_OK;
}
// mean duration (ms) of the sampled launches of the training kernels since the last reset: which = 0 k_step_assign, 1 k_step_update
extern "C" int hamsm_step_kernel_ms(hamsm_ctx *c, int which, double *ms, int64_t *samples, int reset) {
  CTX(c);
  if (which < 0 || which > 2 || !ms) return hamsm_fail(c, HAMSM_EINVAL, "step_kernel_ms: which");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->ks_stamps && !c->ks_pending.empty()) {
    std::vector<unsigned long long> st(2 * c->ks_pending.size());
    HIPCHK(c, hipMemcpy(st.data(), c->ks_stamps, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < c->ks_pending.size(); i++)
      if (st[2 * i + 1] > st[2 * i]) {  // (a launch that only drained a stopped chain leaves no stamps: not an E-step, not counted)
        c->ks_dev_sum_ms += (double)(st[2 * i + 1] - st[2 * i]) * 1e-5;  // 10 ns ticks
        c->ks_dev_n++;
      }
  }
  for (auto &k : c->ks_pending) {
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, k.e0, k.e1) == hipSuccess && hipEventElapsedTime(&b, k.e1, k.e2) == hipSuccess) {
      c->ks_sum_ms[0] += a;
      c->ks_sum_ms[1] += b;
      c->ks_n[0]++;
      c->ks_n[1]++;
    }
    c->ks_free.push_back(k.e0);
    c->ks_free.push_back(k.e1);
    c->ks_free.push_back(k.e2);
  }
  c->ks_pending.clear();
  // (re)arm the stamp slots: sampling is on from the first call of this function on
  if (!c->ks_stamps) HIPCHK(c, hipMalloc((void **)&c->ks_stamps, sizeof(unsigned long long) * 2 * HAMSM_KS_SLOTS));
  {
    std::vector<unsigned long long> init(2 * HAMSM_KS_SLOTS);
    for (int i = 0; i < HAMSM_KS_SLOTS; i++) init[2 * i] = ~0ull, init[2 * i + 1] = 0ull;
    HIPCHK(c, hipMemcpy(c->ks_stamps, init.data(), sizeof(unsigned long long) * init.size(), hipMemcpyHostToDevice));
  }
  // which = 2: the E-step's duration by the device clock (first wave in -> last wave out), without the launch boundary
  if (which == 2) *ms = c->ks_dev_n ? c->ks_dev_sum_ms / (double)c->ks_dev_n : 0.0;
  else *ms = c->ks_n[which] ? c->ks_sum_ms[which] / (double)c->ks_n[which] : 0.0;
  if (samples) *samples = which == 2 ? c->ks_dev_n : c->ks_n[which];
  if (reset) c->ks_sum_ms[0] = c->ks_sum_ms[1] = c->ks_dev_sum_ms = 0.0, c->ks_n[0] = c->ks_n[1] = c->ks_dev_n = 0;
  return HAMSM_OK;
}

extern "C" int hamsm_last_kernel_ms(hamsm_ctx *c, int stage, double *ms) {
  CTX(c);
  if (stage < 0 || stage >= ST_N) return hamsm_fail(c, HAMSM_EINVAL, "bad stage");
  timers_resolve(c, true);
  *ms = c->last_ms[stage];
  return HAMSM_OK;
}

// ---- model state --------------------------------------------------------------------------------
extern "C" int hamsm_set_geometry(hamsm_ctx *c, int D, int ndim, int S) {
  CTX(c);
  if (D < 1 || ndim < 1 || ndim > HAMSM_MAX_NDIM || S < 1 || S > HAMSM_MAX_STRATA)
    return hamsm_fail(c, HAMSM_EINVAL, "bad geometry D=%d ndim=%d S=%d", D, ndim, S);
  if (D > HAMSM_MAX_D) return hamsm_fail(c, HAMSM_ENOTSUP, "D=%d features: at most %d are supported", D, HAMSM_MAX_D);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->filter_skip = 0;  // a new model: the ranking pass gets its chance again
  c->filter_lite_skip = 0;
  c->filter_cnt2 = nullptr;
  if (S != c->S) {
    hipHostFree(c->h_und);
    c->h_und = nullptr;
  }
  if (D == c->D && ndim == c->ndim && S == c->S && (int)c->st.size() == S && c->d_remap && c->d_table) {
    // same geometry again (every cluster_coordinates call configures the context): keep the allocations, forget the
    // model -- unfitted strata, identity remap, mapper to be set anew
    for (auto &t : c->st) t.K = 0;
    std::vector<int32_t> id(S);
    for (int s = 0; s < S; s++) id[s] = s;
    HIPCHK(c, hipMemcpy(c->d_remap, id.data(), sizeof(int32_t) * S, hipMemcpyHostToDevice));
    c->Kfit = 0;
    c->table_dirty = true;
    c->have_mapper = false;
    return HAMSM_OK;
  }
  free_strata(c);
  c->D = D;
  // k-steps of 4 features; rounded up to a value the kernels are instantiated for (zero padded).  More than 128
  // features: KS = ceil(D / 4) exactly, a run-time value of the generic kernels (k_assign_generic, k_step_*_generic)
  {
    static const int ks_list[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    int need = (D + 3) / 4;
    c->KS = need;
    for (int v : ks_list)
      if (v >= need) {
        c->KS = v;
        break;
      }
  }
  c->ndim = ndim;
  c->S = S;
  c->st.resize(S);
  HIPCHK(c, hipMalloc((void **)&c->d_remap, sizeof(int32_t) * S));
  std::vector<int32_t> id(S);
  for (int s = 0; s < S; s++) id[s] = s;  // stratified_clustering.py:85
  HIPCHK(c, hipMemcpy(c->d_remap, id.data(), sizeof(int32_t) * S, hipMemcpyHostToDevice));
  HIPCHK(c, hipMalloc((void **)&c->d_table, sizeof(StratumDev) * S));
  c->mapper.ndim = ndim;
  c->mapper.remap = c->d_remap;
  c->table_dirty = true;
  c->have_mapper = false;
  return HAMSM_OK;
}

extern "C" int hamsm_set_mapper_rectilinear(hamsm_ctx *c, const int32_t *nb, const float *b) {
  CTX(c);
  if (c->st.empty()) return hamsm_fail(c, HAMSM_EINVAL, "set_geometry first");
  long nbins = 1;
  int tot = 0;
  for (int d = 0; d < c->ndim; d++) {
    if (nb[d] < 2) return hamsm_fail(c, HAMSM_EINVAL, "need >=2 boundaries per dim");
    c->mapper.nb[d] = nb[d];
    c->mapper.boff[d] = tot;
    tot += nb[d];
    nbins *= nb[d] - 1;
  }
  if (nbins != c->S) return hamsm_fail(c, HAMSM_EINVAL, "mapper has %ld bins, geometry says %d", nbins, c->S);
  c->mapper.kind = MAPPER_RECT;
  c->ext_bins_set = false;
  // the same boundaries again (every cluster_coordinates call sets the mapper): nothing to upload
  if (c->d_bounds && (int)c->h_bounds.size() == tot && memcmp(c->h_bounds.data(), b, sizeof(float) * tot) == 0) {
    c->mapper.bounds = c->d_bounds;
    c->have_mapper = true;
    return HAMSM_OK;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(c->d_bounds);
  c->d_bounds = nullptr;
  c->mapper.bounds = nullptr;
  c->have_mapper = false;
  c->h_bounds.clear();
  HIPCHK(c, hipMalloc((void **)&c->d_bounds, sizeof(float) * tot));
  HIPCHK(c, hipMemcpy(c->d_bounds, b, sizeof(float) * tot, hipMemcpyHostToDevice));
  c->h_bounds.assign(b, b + tot);
  c->mapper.bounds = c->d_bounds;
  c->have_mapper = true;
  return HAMSM_OK;
}

extern "C" int hamsm_set_mapper_voronoi(hamsm_ctx *c, int n_centers, const float *h_centers) {
  CTX(c);
  if (c->st.empty()) return hamsm_fail(c, HAMSM_EINVAL, "set_geometry first");
  if (n_centers != c->S) return hamsm_fail(c, HAMSM_EINVAL, "mapper has %d centres, geometry says %d bins", n_centers, c->S);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(c->d_centers);
  c->d_centers = nullptr;
  c->have_mapper = false;
  const size_t bytes = sizeof(float) * (size_t)n_centers * c->ndim;
  HIPCHK(c, hipMalloc((void **)&c->d_centers, bytes));
  HIPCHK(c, hipMemcpy(c->d_centers, h_centers, bytes, hipMemcpyHostToDevice));
  c->mapper.kind = MAPPER_VORONOI;
  c->mapper.ncent = n_centers;
  c->mapper.centers = c->d_centers;
  c->ext_bins_set = false;
  c->have_mapper = true;
  return HAMSM_OK;
}

extern "C" int hamsm_set_mapper_external(hamsm_ctx *c) {
  CTX(c);
  if (c->st.empty()) return hamsm_fail(c, HAMSM_EINVAL, "set_geometry first");
  c->mapper.kind = MAPPER_EXTERNAL;
  c->mapper.ext_bins = nullptr;
  c->ext_bins_set = false;
  c->have_mapper = true;
  return HAMSM_OK;
}

extern "C" int hamsm_set_frame_bins(hamsm_ctx *c, const int32_t *d_bins) {
  CTX(c);
  if (c->mapper.kind != MAPPER_EXTERNAL) return hamsm_fail(c, HAMSM_EINVAL, "frame bins are for the external mapper only");
  c->mapper.ext_bins = d_bins;
  c->ext_bins_set = d_bins != nullptr;
  return HAMSM_OK;
}

extern "C" int hamsm_set_frame_bins_host(hamsm_ctx *c, const int32_t *h_bins, int64_t n) {
  CTX(c);
  if (c->mapper.kind != MAPPER_EXTERNAL) return hamsm_fail(c, HAMSM_EINVAL, "frame bins are for the external mapper only");
  if (n <= 0 || !h_bins) return hamsm_fail(c, HAMSM_EINVAL, "frame bins missing");
  const size_t bytes = sizeof(int32_t) * (size_t)n;
  if (bytes > c->ext_bins_own_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(c->d_ext_bins_own);
    c->d_ext_bins_own = nullptr;
    c->ext_bins_own_cap = 0;
    HIPCHK(c, hipMalloc((void **)&c->d_ext_bins_own, bytes + bytes / 4 + 256));
    c->ext_bins_own_cap = bytes + bytes / 4 + 256;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_ext_bins_own, h_bins, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->mapper.ext_bins = c->d_ext_bins_own;
  c->ext_bins_set = true;
  return HAMSM_OK;
}

// external mapper: a call that maps frames to bins needs the bins named for it, and uses them up
int hamsm_take_frame_bins(hamsm_ctx *c, const char *what) {
  if (c->mapper.kind != MAPPER_EXTERNAL) return HAMSM_OK;
  if (!c->ext_bins_set) return hamsm_fail(c, HAMSM_EINVAL, "%s: external mapper -- call hamsm_set_frame_bins first", what);
  c->ext_bins_set = false;
  return HAMSM_OK;
}

extern "C" int hamsm_set_remap(hamsm_ctx *c, const int32_t *r) {
  CTX(c);
  if (c->st.empty()) return hamsm_fail(c, HAMSM_EINVAL, "set_geometry first");
  for (int s = 0; s < c->S; s++)
    if (r[s] < 0 || r[s] >= c->S) return hamsm_fail(c, HAMSM_EINVAL, "remap[%d]=%d out of range", s, r[s]);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->d_remap, r, sizeof(int32_t) * c->S, hipMemcpyHostToDevice));
  return HAMSM_OK;
}

extern "C" int hamsm_set_bounds(hamsm_ctx *c, const double *basis, const double *target) {
  CTX(c);
  if (c->st.empty()) return hamsm_fail(c, HAMSM_EINVAL, "set_geometry first");
  for (int d = 0; d < c->ndim; d++) {
    c->mapper.basis[d][0] = basis[2 * d];
    c->mapper.basis[d][1] = basis[2 * d + 1];
    c->mapper.target[d][0] = target[2 * d];
    c->mapper.target[d][1] = target[2 * d + 1];
  }
  c->have_bounds = true;
  return HAMSM_OK;
}

static inline int kpad(int K) { return (K + 15) & ~15; }

int hamsm_stratum_reserve(hamsm_ctx *c, int s, int K) {
  Stratum &t = c->st[s];
  if (K <= t.Kcap) return HAMSM_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipFree(t.C);
  hipFree(t.csq);
  hipFree(t.counts);
  hipFree(t.Bpack);
  t.C = t.csq = t.counts = t.Bpack = nullptr;  // a failed hipMalloc below must not leave dangling pointers behind
  t.Kcap = 0;
  t.K = 0;
  c->table_dirty = true;
  int cap = kpad(K);
  HIPCHK(c, hipMalloc((void **)&t.C, sizeof(double) * (size_t)cap * c->D));
  HIPCHK(c, hipMalloc((void **)&t.csq, sizeof(double) * cap));
  HIPCHK(c, hipMalloc((void **)&t.counts, sizeof(double) * cap));
  HIPCHK(c, hipMalloc((void **)&t.Bpack, sizeof(double) * (size_t)cap * c->KS * 4));
  t.Kcap = cap;
  return HAMSM_OK;
}

// csq (fma chain unless given) + MFMA B-fragment packing of rows [j0, j1) ... and padding up to Kpad
__global__ void k_pack(const double *__restrict__ C, const double *__restrict__ csq_in, int K, int Kpad, int D, int KS,
                       double *__restrict__ csq, double *__restrict__ Bpack) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Kpad) return;
  const double inf = __builtin_huge_val();
  if (j < K) {
    if (csq_in) csq[j] = csq_in[j];
    else {
      double a = 0.0;
      for (int k = 0; k < D; k++) a = __builtin_fma(C[(size_t)j * D + k], C[(size_t)j * D + k], a);
      csq[j] = a;
    }
  } else csq[j] = inf;
  int jt = j >> 4, col = j & 15;
  for (int k = 0; k < KS * 4; k++) {
    double v = (j < K && k < D) ? C[(size_t)j * D + k] : 0.0;
    Bpack[((size_t)jt * KS + (k >> 2)) * 64 + (k & 3) * 16 + col] = v;
  }
}
// the same for long rows (D > 128): the fragment fill is spread over the row (blockIdx.y: 64 features each); ||c||^2 stays ONE fma
// chain over ascending k per centroid (the contract), formed by the threads of the first slice
__global__ void k_pack_long(const double *__restrict__ C, const double *__restrict__ csq_in, int K, int Kpad, int D, int KS,
                            double *__restrict__ csq, double *__restrict__ Bpack) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, k0 = blockIdx.y * 64;
  if (j >= Kpad) return;
  if (k0 == 0) {
    if (j < K) {
      if (csq_in) csq[j] = csq_in[j];
      else {
        double a = 0.0;
        for (int k = 0; k < D; k++) a = __builtin_fma(C[(size_t)j * D + k], C[(size_t)j * D + k], a);
        csq[j] = a;
      }
    } else csq[j] = __builtin_huge_val();
  }
  const int jt = j >> 4, col = j & 15;
  for (int k = k0; k < k0 + 64 && k < KS * 4; k++) {
    const double v = (j < K && k < D) ? C[(size_t)j * D + k] : 0.0;
    Bpack[((size_t)jt * KS + (k >> 2)) * 64 + (k & 3) * 16 + col] = v;
  }
}

// the strata a seeding call has just seeded (train.hip): counts := 0 (_kmeans.py:2271), csq, fragments -- ONE launch for up to
// PACK_MULTI strata (they used to be a memset and a k_pack launch each, between the seeding kernel and the first training step)
__global__ void k_pack_multi(PackMulti pm, int D, int KS) {
  const int y = blockIdx.y;
  const int K = pm.K[y], Kpad = pm.Kp[y];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Kpad) return;
  const double *C = pm.C[y];
  double *csq = pm.csq[y], *Bpack = pm.Bpack[y];
  if (j < K) {
    pm.counts[y][j] = 0.0;
    double a = 0.0;
    for (int k = 0; k < D; k++) a = __builtin_fma(C[(size_t)j * D + k], C[(size_t)j * D + k], a);
    csq[j] = a;
  } else csq[j] = __builtin_huge_val();
  const int jt = j >> 4, col = j & 15;
  for (int k = 0; k < KS * 4; k++) {
    const double v = (j < K && k < D) ? C[(size_t)j * D + k] : 0.0;
    Bpack[((size_t)jt * KS + (k >> 2)) * 64 + (k & 3) * 16 + col] = v;
  }
}

int hamsm_pack_seeded(hamsm_ctx *c, const int *strata, int n) {
  if (c->KS > 32) {  // long rows: the per-stratum form
    for (int i = 0; i < n; i++) {
      Stratum &t = c->st[strata[i]];
      HIPCHK(c, hipMemsetAsync(t.counts, 0, sizeof(double) * t.K, c->stream));
      int rc = hamsm_pack_stratum(c, strata[i], nullptr);
      if (rc) return rc;
    }
    return HAMSM_OK;
  }
  for (int i0 = 0; i0 < n; i0 += PACK_MULTI) {
    PackMulti pm;
    memset(&pm, 0, sizeof pm);
    const int m = n - i0 < PACK_MULTI ? n - i0 : PACK_MULTI;
    int kpmax = 0, live = 0;
    for (int q = 0; q < m; q++) {
      Stratum &t = c->st[strata[i0 + q]];
      if (t.K == 0) continue;
      pm.C[live] = t.C;
      pm.csq[live] = t.csq;
      pm.Bpack[live] = t.Bpack;
      pm.counts[live] = t.counts;
      pm.K[live] = t.K;
      pm.Kp[live] = kpad(t.K);
      kpmax = pm.Kp[live] > kpmax ? pm.Kp[live] : kpmax;
      live++;
    }
    if (live) hipLaunchKernelGGL(k_pack_multi, dim3((kpmax + 63) / 64, live), dim3(64), 0, c->stream, pm, c->D, c->KS);
  }
  HIPCHK(c, hipGetLastError());
  return HAMSM_OK;
}

int hamsm_pack_stratum(hamsm_ctx *c, int s, const double *d_csq_in) {
  Stratum &t = c->st[s];
  if (t.K == 0) return HAMSM_OK;
  int Kp = kpad(t.K);
  if (c->KS > 32)
    hipLaunchKernelGGL(k_pack_long, dim3((Kp + 63) / 64, (c->KS * 4 + 63) / 64), dim3(64), 0, c->stream, t.C, d_csq_in, t.K, Kp, c->D, c->KS,
                       t.csq, t.Bpack);
  else
    hipLaunchKernelGGL(k_pack, dim3((Kp + 63) / 64), dim3(64), 0, c->stream, t.C, d_csq_in, t.K, Kp, c->D, c->KS, t.csq, t.Bpack);
  HIPCHK(c, hipGetLastError());
  return HAMSM_OK;
}

extern "C" int hamsm_set_centroids(hamsm_ctx *c, int s, int K, const double *C, const double *csq, const double *counts) {
  CTX(c);
  if (c->st.empty() || s < 0 || s >= c->S || K < 0) return hamsm_fail(c, HAMSM_EINVAL, "bad stratum/K");
  HIPCHK(c, hipSetDevice(c->device));
  Stratum &t = c->st[s];
  c->table_dirty = true;
  if (K == 0) {
    t.K = 0;
    return HAMSM_OK;
  }
  int rc = hamsm_stratum_reserve(c, s, K);
  if (rc) return rc;
  t.K = K;
  HIPCHK(c, hipMemcpyAsync(t.C, C, sizeof(double) * (size_t)K * c->D, hipMemcpyHostToDevice, c->stream));
  if (counts) HIPCHK(c, hipMemcpyAsync(t.counts, counts, sizeof(double) * K, hipMemcpyHostToDevice, c->stream));
  else HIPCHK(c, hipMemsetAsync(t.counts, 0, sizeof(double) * K, c->stream));
  double *d_csq_in = nullptr;
  if (csq) {
    rc = hamsm_ensure_scratch(c, sizeof(double) * K);
    if (rc) return rc;
    d_csq_in = (double *)c->scratch;
    HIPCHK(c, hipMemcpyAsync(d_csq_in, csq, sizeof(double) * K, hipMemcpyHostToDevice, c->stream));
  }
  rc = hamsm_pack_stratum(c, s, d_csq_in);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));  // host buffers are the caller's
  return HAMSM_OK;
}

extern "C" int hamsm_get_centroids(hamsm_ctx *c, int s, double *C, double *counts, double *csq) {
  CTX(c);
  if (c->st.empty() || s < 0 || s >= c->S) return hamsm_fail(c, HAMSM_EINVAL, "bad stratum");
  Stratum &t = c->st[s];
  if (t.K == 0) return HAMSM_OK;
  if (C) HIPCHK(c, hipMemcpyAsync(C, t.C, sizeof(double) * (size_t)t.K * c->D, hipMemcpyDeviceToHost, c->stream));
  if (counts) HIPCHK(c, hipMemcpyAsync(counts, t.counts, sizeof(double) * t.K, hipMemcpyDeviceToHost, c->stream));
  if (csq) HIPCHK(c, hipMemcpyAsync(csq, t.csq, sizeof(double) * t.K, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HAMSM_OK;
}

extern "C" int hamsm_num_centroids(hamsm_ctx *c, int s) {
  CTX(c);
  if (c->st.empty() || s < 0 || s >= c->S) return hamsm_fail(c, HAMSM_EINVAL, "bad stratum");
  return c->st[s].K;
}

extern "C" int hamsm_get_counts(hamsm_ctx *c, int s, double *counts) {
  CTX(c);
  if (c->st.empty() || s < 0 || s >= c->S) return hamsm_fail(c, HAMSM_EINVAL, "bad stratum");
  Stratum &t = c->st[s];
  if (t.K == 0) return HAMSM_OK;
  HIPCHK(c, hipMemcpyAsync(counts, t.counts, sizeof(double) * t.K, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HAMSM_OK;
}

int hamsm_refresh_table(hamsm_ctx *c) {
  if (!c->table_dirty) return HAMSM_OK;
  std::vector<StratumDev> tab(c->S);
  int off = 0;
  for (int s = 0; s < c->S; s++) {
    tab[s].Bpack = c->st[s].Bpack;
    tab[s].csq = c->st[s].csq;
    tab[s].K = c->st[s].K;
    tab[s].off = off;  // stratified_clustering.py:178-185
    off += c->st[s].K;
  }
  c->Kfit = off;  // :143-150
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->d_table, tab.data(), sizeof(StratumDev) * c->S, hipMemcpyHostToDevice));
  c->table_dirty = false;
  return HAMSM_OK;
}

// common.h -- context, error handling and device-side helpers shared by the HIP translation units.
// gfx950 (CDNA4, wave64) only.  Built with -ffp-contract=off: a multiply and an add are never
// fused unless the source says __builtin_fma, which is how the arithmetic contract of
// include/hamsm.h (and of oracle/hamsm_oracle.c) is kept bit-for-bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/hamsm.h"

#define HAMSM_MAX_NDIM 8
#define HAMSM_MAX_D 16384  // features per frame (more than 128: the generic kernels)
#define HAMSM_MAX_STRATA 4096
#define HAMSM_KS_SLOTS 1024  // sampled training launches between two reads of hamsm_step_kernel_ms
#define POT_LANES 1024  // strided partial sums of k-means++ potentials (oracle: POT_LANES, POT_GROUP = 32)
#define SCAN_CHUNK 64   // chunked cumulative sum (oracle: SCAN_CHUNK)

// ---- device view of the mapper / bounds (passed by value to kernels) ---------------------------
#define MAPPER_RECT 0
#define MAPPER_VORONOI 1
#define MAPPER_EXTERNAL 2
struct MapperDev {
  int kind;                  // MAPPER_*
  int ncent;                 // Voronoi: centres (= strata)
  const float *centers;      // Voronoi: device [ncent][ndim]
  const int32_t *ext_bins;   // external: bin of frame i of the current call (device)
  int ndim;
  int nb[HAMSM_MAX_NDIM];    // boundaries per dim
  int boff[HAMSM_MAX_NDIM];  // offset of each dim's boundaries in `bounds`
  const float *bounds;       // device
  const int32_t *remap;      // device [S]
  double basis[HAMSM_MAX_NDIM][2];
  double target[HAMSM_MAX_NDIM][2];
};

// ---- per-stratum centroid storage ---------------------------------------------------------------
// C      [Kcap][D]              row-major f64 (update / get)
// csq    [Kpad]                 ||c||^2, +inf beyond K (never selected)
// counts [Kcap]
// Bpack  [Kpad/16][KS][64]      MFMA B-fragment order: element (jt, ks, lane) = C[jt*16 + (lane&15)][ks*4 + (lane>>4)]
//                               (v_mfma_f64_16x16x4_f64: lane l holds B[k = l>>4][col = l&15]); zero padded
struct StratumDev {
  const double *Bpack;
  const double *csq;
  int K;    // fitted centroids (0 = unfitted)
  int off;  // label offset = sum of K of earlier strata
};

struct Stratum {
  int K = 0, Kcap = 0;
  double *C = nullptr, *csq = nullptr, *counts = nullptr, *Bpack = nullptr;
};

enum { ST_ASSIGN = 0, ST_FLUX, ST_STATS, ST_SEED, ST_STEP, ST_REASSIGN, ST_PREPASS, ST_OTHER, ST_N };

#define HAMSM_CTX_LIVE 0x68614d534d686970ull  // "haMSMhip"
#define HAMSM_CTX_DEAD 0xdeadc0de0badc7c7ull

struct hamsm_ctx {
  uint64_t magic = 0;  // HAMSM_CTX_LIVE between hamsm_create and hamsm_destroy (the struct is never returned to the heap)
  char err[512];
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // uploads that overlap host work (hamsm_memcpy_h2d_async)
  std::unordered_map<void *, size_t> host_live;  // page-locked blocks handed out by hamsm_malloc_host
  // training steps of different strata are independent chains of small latency-bound kernels: strata alternate
  // between `stream` and these side queues so that the chains overlap (fork/join with events inside run_steps)
  hipStream_t side_stream[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  // bounded run-ahead of the training enqueue (run_steps, MFMA-bound calls): a ring of events per queue, a queue of its own and a
  // page-locked buffer for reading the chains' stop words while the pass runs
  hipEvent_t ev_ring[4][4] = {};
  hipStream_t poll_stream = nullptr;
  int32_t *h_poll = nullptr;
  size_t h_poll_n = 0;
  int step_lanes = 2;  // queues used by the training steps (HAMSM_STEP_LANES = 1..4)
#ifdef HAMSM_EXPERIMENTS
  // ---- csrc/experiments/ (libhamsm_hip_exp.so only; every one of these measured slower than the shipped path and ships OFF) ----
  int train_persist = 0;  // HAMSM_TRAIN_PERSIST=1: planned training passes run as ONE persistent launch (measured slower than the
                          // multi-kernel path on every shape tried -- DESIGN.md §5.3a -- and therefore off; kept as a tested alternative)
  int persist_wpx = 64;   // workgroups per XCD of that launch (HAMSM_PERSIST_WPX, 8 XCDs: grid = 8 x this)
  // pipelined training (train.hip; an experiment, OFF by default -- HAMSM_TRAIN_AHEAD=1): ONE launch per WE
  // iteration -- the M-step of iteration i beside the E-step of iteration i + 2, which runs against centroids two updates old.
  // A label is final only where a displacement bound proves that no update can change its argmin; the other rows are decided
  // exactly, one step ahead, by the waves that own the candidate centroids; every M-step checks the bound against the
  // displacement really seen.  Bit-identical to the step-by-step path; measured slower on the bench's data.
  int ahead = 0;
  int ahead_n0 = 200;       // iterations of a pass that run step by step before the first segment (HAMSM_AHEAD_N0)
  int ahead_gmax = 16;      // segments hold at most 64 x this many iterations (HAMSM_AHEAD_G)
  int ahead_rt = 1;         // 16-row tiles per workgroup of the pipelined E-step (HAMSM_AHEAD_RT = 1 | 2; 2 measured slower)
  double ahead_alpha = 3.0; // displacement allowed = alpha x the latest measurement (HAMSM_AHEAD_ALPHA)
  long long ahead_stat[6] = {0, 0, 0, 0, 0, 0};  // segments, steps inside them, premises failed, -, relaunch passes, steps run step by step
  // (experiment, OFF: measured slower) a training call whose E-step is bound by the f64 matrix unit
  // (>= ranked_min_flop per launch group) can run it through the assignment kernels (ranking pass + exact kernel: same labels;
  // train.hip run_steps).  HAMSM_TRAIN_RANKED=1: by size, 2: always (tests), 0: never (default)
  int train_ranked = 0;
  double ranked_min_flop = 1.0e9;  // (HAMSM_RANKED_MIN_MFLOP)
  long long ranked_groups = 0;     // launch groups served that way (debug counter)
  int train_graph = 0;      // HAMSM_TRAIN_GRAPH=1 (experiment): a training pass's launches captured into one hipGraph
  bool capturing = false;   // ... while its stream capture is open (no event timing inside)
#endif
  // the E-step of training calls that are bound by the f64 matrix unit ranks the centroids on the bf16 unit (k_step_assign_ranked,
  // train.hip; same labels): 0 never, 1 when a run's heaviest call has >= step_ranked_min_flop (default), 2 always
  int step_ranked = 1;                      // HAMSM_STEP_RANKED
  int step_direct_plain = 1;                // HAMSM_STEP_DIRECT_PLAIN=0: the f64-MFMA training kernels WITH the f64 batch copy (round 4's form)
  int enqueue_threads = 2;                  // HAMSM_ENQUEUE_THREADS=1: one host thread feeds both queues of a latency-bound training pass
  int step_lrows = 1;                       // HAMSM_STEP_LROWS: ... with the call's row list in LDS (batches of <= 4096 rows)
  bool cur_direct = false;                  // (the run being enqueued)
  int flux_last_path = -1;                  // (test hook) 0 atomics, 1 sorted, 2 by rows
  int flux_rows = 0, flux_rows_max_r = 5400;  // HAMSM_FLUX_ROWS / HAMSM_FLUX_ROWS_MAX_R: deterministic flux by rows (flux_rows.hip) up to this many states
  int step_pipe = 1;                        // HAMSM_STEP_PIPE: ranked E-step sweep as a software pipeline (MFMAs of the next unit between the ranking of this one)
  int step_wide = 0;                        // HAMSM_STEP_WIDE: 256-row workgroups in the ranked E-step where 128-row ones overfill the chip
  int n_cu = 256;                           // compute units of the device (hamsm_create)
  int step_direct = 1;                      // HAMSM_STEP_DIRECT: ranked runs: the M-step fetches member rows from the frames through the
                                            // row list (no f64 batch copy written by the E-step and read back); 0 = round 4's copy
  double step_ranked_min_flop = 1.5e8;      // HAMSM_STEP_RANKED_MFLOP: flop of a run's heaviest call from which the ranked E-step is taken
  unsigned long long rank_stat[4] = {0, 0, 0, 0};  // rows certified / settled between two candidates / swept exactly; runs
  int generic_stage_kb = 150;  // D > 128: the E-step stages its 16 rows in LDS while they fit this (HAMSM_GENERIC_STAGE_KB; 0 = never)
  // hamsm_malloc / hamsm_free keep released blocks for reuse (hipMalloc costs ~1 ms per 40 MB block and hipFree
  // synchronises the device; every use of a block is ordered on `stream`, so reuse needs no synchronisation)
  std::unordered_map<void *, size_t> blk_live;
  std::multimap<size_t, void *> blk_cache;
  size_t blk_cached_bytes = 0;
  int D = 0, KS = 0 /* ceil(D/4) */, ndim = 0, S = 0;
  bool have_mapper = false, have_bounds = false;
  MapperDev mapper;  // host copy with device pointers inside
  float *d_bounds = nullptr;
  float *d_centers = nullptr;    // Voronoi centres
  int32_t *d_ext_bins_own = nullptr;  // device copy of host-supplied frame bins (hamsm_set_frame_bins_host)
  size_t ext_bins_own_cap = 0;
  bool ext_bins_set = false;     // external mapper: bins named for the next call
  std::vector<float> h_bounds;  // host copy of what d_bounds holds
  int32_t *d_remap = nullptr;
  std::vector<Stratum> st;
  StratumDev *d_table = nullptr;  // [S]
  bool table_dirty = true;
  int Kfit = 0;
  int *d_err = nullptr;   // device error word: [0] code, [1] detail a, [2] detail b
  int *h_err = nullptr;   // pinned mirror
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
  void *scratch2 = nullptr;  // second arena: internals of hamsm_assign (its caller may hold `scratch`)
  size_t scratch2_bytes = 0;
  double *flux_part = nullptr;  // per-XCD partial flux matrices (flux.hip)
  size_t flux_part_bytes = 0;
  int flux_xcd = 1;             // HAMSM_FLUX_XCD=0: one matrix, device-scope atomics
  double *pca_S2 = nullptr, *pca_mu0 = nullptr;  // streaming PCA fit (pca.hip): running sums [Dp][Dp], shift [Dp]
  int pca_D = 0;
  int64_t pca_n = 0;
  int flux_xcd_checked = 0;     // the per-XCD path has been compared with the single-matrix kernel once (flux.hip)
  std::vector<char> h_stage;  // host staging for small asynchronous H2D copies (lives as long as the ctx)
  void *h_pinned = nullptr;  // small pinned staging buffer
  size_t h_pinned_bytes = 0;
  struct EvPair {
    hipEvent_t e0, e1;
    int stage;
  };
  std::vector<EvPair> ev_pending;
  std::vector<hipEvent_t> ev_free;
  hipEvent_t ev_cur0 = nullptr;
  double timers[ST_N] = {0};
  double last_ms[ST_N] = {0};
  // sampled per-launch durations of the two training kernels (every 64th launch group of queue 0: events around the E-step
  // and the M-step launch), for the bench line's roofline of the dominant kernel
  struct KSample {
    hipEvent_t e0, e1, e2;
  };
  std::vector<KSample> ks_pending;
  std::vector<hipEvent_t> ks_free;  // recycled events of resolved samples
  unsigned long long *ks_stamps = nullptr;  // device [4096][2]: first-wave-in / last-wave-out stamps of the sampled E-step launches
  double ks_dev_sum_ms = 0.0;               // ... summed kernel-only durations (100 MHz clock)
  double ks_sum_ms[2] = {0, 0};
  long long ks_n[2] = {0, 0};
  long long ks_dev_n = 0;                   // ... and how many of them carried both stamps
  long long ks_tick = 0;
  size_t seed_g_budget = (size_t)4 << 30;  // bytes of Gram matrices resident at once (HAMSM_SEED_G_MB): 56 strata of 3072 rows
  unsigned long long *h_und = nullptr;  // pinned [2][S]: undecided counts of the last filtered assign (slot 1: the child side of a pair)
  int und_slot = 0;                     // slot the assign being enqueued reports into
  long long exact_cap = 0;              // workgroups the exact pass behind the ranking pass is launched with (0: the full upper bound)
  long long exact_cap_of[2] = {0, 0};   // ... per slot, for the check after the synchronisation
  int exact_frb_of[2] = {0, 0};         // frames per workgroup of that launch
  bool exact_full = false;              // (re-run of a side whose undecided frames outgrew the capped grid)
  int exact_capped = 1;                 // HAMSM_EXACT_CAPPED=0: the exact pass behind the ranking pass as before (512-frame workgroups, full grid)
  long long exact_overflows = 0;        // (test hook) sides run again because their undecided frames outgrew the capped grid
  int filter_skip = 0;     // calls left that bypass the ranking pass (it settled < 3/4 of the frames last time)
  int filter_ran = 0;      // the last assign ran the ranking pass
  int filter_bf16 = 1;     // bf16 x 3 form of the ranking pass where it applies (D <= 32); HAMSM_FILTER_BF16=0: f32 form
  int filter_nb = 8;       // most frame blocks per filter workgroup (HAMSM_FILTER_NB, <= 16)
  int filter_lite = 1;     // bf16 form: 3 of the 6 products first (x1c1, x1c2, x2c1: 2^-14-level terms dropped, bound 4x wider); steps
                           // aside for 7 calls when it settles < 31/32 of the frames (HAMSM_FILTER_LITE=0: never, =2: always)
  int filter_lite_skip = 0;
  int filter_dbuf = 1;     // bf16 form with fragment groups: double-buffered LDS filled by global_load_lds (HAMSM_FILTER_DBUF=0: single buffer)
  int filter_form = 0;     // what the last assign ran: 0 exact only, 1 f32, 2 bf16 x 3 (6 products), 3 bf16 lite (3 products)
  int filter_lds_kb = 72;  // LDS budget of a filter workgroup (HAMSM_FILTER_LDS_KB)
  int assign_filter = 1;   // f32-MFMA filter in front of the exact f64 distance kernel (HAMSM_ASSIGN_FILTER=0: exact kernel only)
  unsigned long long *filter_cnt2 = nullptr;  // (debug) uncertain-row counters of the last filtered assign, [S] in scratch2
  int assign_waves = 4;    // waves per workgroup of the sorted assign kernel (HAMSM_ASSIGN_W=8: 8 waves x 2 centroid tiles)
  unsigned long long scratch_epoch = 0, seed_chosen_epoch = 0;  // arena generation; ... of the seeding whose choices are recorded below
  void *seed_chosen_dev = nullptr;  // chosen indices of the last seeding's first item (device)
  std::vector<std::pair<void *, int>> seed_chosen_items;  // ... of every item of the last seeding call (device pointer, K)
  void *comm = nullptr;  // ncclComm_t
  void *comm_buf = nullptr;  // device words of the collectives' argument / readiness exchanges (allocated with the communicator)
  size_t comm_buf_bytes = 0;
  int nranks = 1, rank = 0;
};

int hamsm_fail(hamsm_ctx *c, int code, const char *fmt, ...);
// every entry point starts with CTX(c): the handle must be one hamsm_create returned and hamsm_destroy has not seen
// (looked up in a registry before it is dereferenced: a stale or foreign handle is an error, never a wild access)
int hamsm_ctx_check(hamsm_ctx *c, const char *fn);
#define CTX(c)                                   \
  do {                                           \
    int rc_ = hamsm_ctx_check((c), __func__);    \
    if (rc_) return rc_;                         \
  } while (0)

#define HIPCHK(c, call)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return hamsm_fail((c), HAMSM_EDEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

void hamsm_pca_release(hamsm_ctx *c);
int hamsm_take_frame_bins(hamsm_ctx *c, const char *what);  // external mapper: consume the bins named for this call
int hamsm_ensure_scratch(hamsm_ctx *c, size_t bytes);
int hamsm_ensure_scratch2(hamsm_ctx *c, size_t bytes);
int hamsm_ensure_pinned(hamsm_ctx *c, size_t bytes);
int hamsm_refresh_table(hamsm_ctx *c);                    // rebuilds d_table / Kfit when dirty
int hamsm_check_device_error(hamsm_ctx *c, const char *what);  // reads d_err (syncs the stream)
int hamsm_stratum_reserve(hamsm_ctx *c, int s, int K);
int hamsm_pack_stratum(hamsm_ctx *c, int s, const double *d_csq_in);
#define PACK_MULTI 8
struct PackMulti {
  const double *C[PACK_MULTI];
  double *csq[PACK_MULTI], *Bpack[PACK_MULTI], *counts[PACK_MULTI];
  int K[PACK_MULTI], Kp[PACK_MULTI];
};
int hamsm_pack_seeded(hamsm_ctx *c, const int *strata, int n);
#ifdef HAMSM_EXPERIMENTS
int hamsm_assign_rows(hamsm_ctx *c, const void *X, int dt, int64_t ldx, int64_t n_span, int32_t *d_labels, const int32_t *d_perm,
                      const unsigned long long *d_cnt);  // assign.hip: the E-step of a heavy training call (experiments/train_ranked.inc)
#define HAMSM_CAPTURING(c) ((c)->capturing)
#else
#define HAMSM_CAPTURING(c) false
#endif
void hamsm_timer_begin(hamsm_ctx *c);
void hamsm_timer_end(hamsm_ctx *c, int stage);

// ---- device helpers -----------------------------------------------------------------------------
#ifdef __HIPCC__
// westpa RectilinearBinMapper.assign restated (oracle bin_of): float32 compare, half-open bins,
// row-major over dims; -1 outside the bin space (also NaN).
__device__ __forceinline__ int dev_bin_of(const MapperDev &m, const double *p, int64_t i) {
  if (m.kind == MAPPER_EXTERNAL) return m.ext_bins[i];  // the mapper's own `assign` ran on the host
  if (m.kind == MAPPER_VORONOI) {  // nearest centre, float32 coordinates, float64 squared distance, first minimum
    int best = -1;
    double bd = 0.0;
    for (int c = 0; c < m.ncent; c++) {
      double d2 = 0.0;
      for (int d = 0; d < m.ndim; d++) {
        const double t = (double)(float)p[d] - (double)m.centers[c * m.ndim + d];
        d2 = d2 + t * t;
      }
      if (best < 0 || d2 < bd) {
        bd = d2;
        best = c;
      }
    }
    return bd == bd ? best : -1;  // NaN coordinates are outside the bin space
  }
  int idx = 0;
  for (int d = 0; d < m.ndim; d++) {
    float pf = (float)p[d];
    const float *b = m.bounds + m.boff[d];
    int n = m.nb[d], lo = 0, hi = n;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (b[mid] <= pf) lo = mid + 1;
      else hi = mid;
    }
    int j = lo - 1;
    if (j < 0 || j >= n - 1) return -1;
    idx = idx * (n - 1) + j;
  }
  return idx;
}

// The same for the rectilinear mapper with its boundaries staged in LDS by the caller (lb: all dims' boundaries, as `bounds`): the
// bisection is a chain of dependent loads per frame and dimension, from LDS instead of L2.  nullptr: dev_bin_of.
__device__ __forceinline__ int dev_bin_of_staged(const MapperDev &m, const float *lb, const double *p, int64_t i) {
  if (!lb) return dev_bin_of(m, p, i);
  int idx = 0;
  for (int d = 0; d < m.ndim; d++) {
    const float pf = (float)p[d];
    const float *b = lb + m.boff[d];
    int n = m.nb[d], lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (b[mid] <= pf) lo = mid + 1;
      else hi = mid;
    }
    const int j = lo - 1;
    if (j < 0 || j >= n - 1) return -1;
    idx = idx * (n - 1) + j;
  }
  return idx;
}
// boundaries of a rectilinear mapper that a kernel may stage (0: none -- another mapper kind, or too many)
static inline int hamsm_mapper_stage_count(const MapperDev &m) {
  if (m.kind != MAPPER_RECT) return 0;
  const int nbt = m.boff[m.ndim - 1] + m.nb[m.ndim - 1];
  return nbt <= 4096 ? nbt : 0;
}

// msm_we.py:480-492 / 515-527: all dims strictly inside, f64
__device__ __forceinline__ bool dev_in_box(const double (*box)[2], int ndim, const double *p) {
  bool in = true;
  for (int d = 0; d < ndim; d++) in = in && (p[d] > box[d][0]) && (p[d] < box[d][1]);
  return in;
}

template <typename XT>
__device__ __forceinline__ double dev_ld(const XT *X, int64_t i) {
  return (double)X[i];
}

__device__ __forceinline__ void dev_set_error(int *err, int code, int a, int b) {
  if (atomicCAS(&err[0], 0, code) == 0) {
    err[1] = a;
    err[2] = b;
  }
}
#endif

/* hamsm_oracle.c -- CPU restatement of the reference hot path, exporting the ABI of include/hamsm.h.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the `cpu_baseline` "port": only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  Nothing under msm_we_amd/
 * references it; the product library is msm_we_amd/csrc (HIP, gfx950).
 *
 * Pinning: checked against golden vectors produced by the real reference
 * (jdrusso/msm_we 0.1.28.dev1 + scikit-learn 1.7.2) in tests/golden/ (.npz files) -- labels identical,
 * centroids to 1e-12, flux/T/pSS/J to 1e-6 -- and against the reference's own NTL9 fixtures
 * (tests/golden/ntl9/, .npy files) for the transition-matrix stage.
 *
 * Arithmetic contract shared with the HIP library (bit-for-bit):
 *  - dot(x, c) = fma chain over ascending feature index from +0.0 (what OpenBLAS dgemm computes
 *    per element for K <= GEMM_Q; verified in this container for sklearn's single-row predict);
 *    distance key = csq[j] - 2*dot (one rounding); first minimum wins (strict <).
 *  - everything else is separate multiply / add (no contraction): build with -ffp-contract=off.
 *  - reductions whose order the reference leaves to BLAS/NumPy SIMD (k-means++ potentials, row
 *    norms) use the fixed, GPU-reproducible orders documented at each function.
 */
#include "../include/hamsm.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#include <errno.h>

#define MAX_NDIM 8
#define POT_LANES 1024 /* strided partial sums for potentials */
#define POT_GROUP 32   /* lanes per second-level group        */
#define SCAN_CHUNK 64 /* chunked cumulative sum              */

typedef struct {
  int K;          /* fitted centroids, 0 = unfitted */
  double *C;      /* [K][D] */
  double *csq;    /* [K]    */
  double *counts; /* [K]    */
} stratum_t;

#define CTX_MAGIC_LIVE 0x68614d534d6f7263ull /* "haMSMorc" */
#define CTX_MAGIC_DEAD 0xdeadc0de0badc7c7ull

struct hamsm_ctx {
  uint64_t magic;   /* CTX_MAGIC_LIVE between hamsm_create and hamsm_destroy */
  struct hamsm_ctx *next; /* registry of every ctx ever created (see ctx_check) */
  void **blocks;    /* live hamsm_malloc blocks of this ctx */
  size_t n_blocks, cap_blocks;
  char err[512];
  int D, ndim, S;
  int nb[MAX_NDIM];
  float *bounds[MAX_NDIM];
  int32_t *remap;
  double basis[MAX_NDIM][2], target[MAX_NDIM][2];
  int have_bounds, have_mapper;
  int mapper_kind;          /* 0 rectilinear, 1 Voronoi (Euclidean), 2 external (bins supplied per call) */
  float *centers;           /* Voronoi [S][ndim] */
  const int32_t *ext_bins;  /* external: bins of the next call's frames */
  int32_t *ext_own;         /* copy of host-supplied bins */
  int ext_set;
  stratum_t *st;
  int64_t *seed_chosen;     /* chosen batch indices of the last hamsm_kmeans_seed_multi call, items back to back */
  int64_t *seed_chosen_off; /* [n_items + 1] */
  int seed_chosen_n;
  double *pca_S2, *pca_mu0; /* streaming PCA fit: running sums [Dp][Dp] (a <= b), shift [Dp] */
  int pca_D;
  int64_t pca_n;
};

static int fail(hamsm_ctx *c, int code, const char *fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
  }
  return code;
}

int hamsm_abi_version(void) { return HAMSM_ABI_VERSION; }
const char *hamsm_backend_name(void) { return "cpu-oracle"; }

/* Handle hygiene (round-2 fix).  A ctx struct is NEVER returned to the heap: hamsm_destroy releases what
 * the ctx owns, stamps it CTX_MAGIC_DEAD and leaves the struct (one private page) on a registry list, so that a stale handle (a wrapper
 * that outlived its context, a second destroy) is recognised for ever instead of aliasing whatever malloc placed there
 * next -- which is how a live context once came to read as zero-filled.  Every entry point starts with CTX(c): the
 * handle must be on the registry (checked WITHOUT dereferencing it first) and live. */
static hamsm_ctx *g_registry = NULL;
static char g_stale_msg[128];

static int ctx_known(const hamsm_ctx *c) {
  for (const hamsm_ctx *q = g_registry; q; q = q->next)
    if (q == c) return 1;
  return 0;
}
static int ctx_check(hamsm_ctx *c, const char *fn) {
  if (!c) return HAMSM_EINVAL;
  if (!ctx_known(c)) {
    snprintf(g_stale_msg, sizeof g_stale_msg, "%s: %p is not a hamsm context of this library", fn, (void *)c);
    return HAMSM_EINVAL;
  }
  if (c->magic == CTX_MAGIC_DEAD) {
    snprintf(g_stale_msg, sizeof g_stale_msg, "%s: stale handle %p (context was destroyed)", fn, (void *)c);
    return HAMSM_EINVAL;
  }
  if (c->magic != CTX_MAGIC_LIVE) { /* neither live nor destroyed: somebody else wrote here -- say so, with evidence */
    const unsigned char *b = (const unsigned char *)c;
    size_t nz = 0, first = sizeof *c;
    for (size_t i = 0; i < sizeof *c; i++)
      if (b[i]) {
        nz++;
        if (first == sizeof *c) first = i;
      }
    snprintf(g_stale_msg, sizeof g_stale_msg, "%s: context %p OVERWRITTEN (magic %llx, %zu of %zu bytes non-zero, first at %zu)",
             fn, (void *)c, (unsigned long long)c->magic, nz, sizeof *c, first);
    return HAMSM_EINVAL;
  }
  return HAMSM_OK;
}
#define CTX(c)                                  \
  do {                                          \
    int rc_ = ctx_check((c), __func__);         \
    if (rc_) return rc_;                        \
  } while (0)

int hamsm_create(int device, hamsm_ctx **out) {
  (void)device;
  if (!out) return HAMSM_EINVAL;
  /* the struct lives on pages of its own, between two inaccessible guard pages: it shares no page and no allocator bin
   * with NumPy buffers, the HIP runtime's host objects or anything that library pins for DMA */
  const size_t pg = 4096, body = (sizeof(hamsm_ctx) + pg - 1) / pg * pg;
  char *m = (char *)mmap(NULL, body + 2 * pg, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (m == MAP_FAILED) return HAMSM_ENOMEM;
  if (mprotect(m + pg, body, PROT_READ | PROT_WRITE) != 0) return HAMSM_ENOMEM;
  hamsm_ctx *c = (hamsm_ctx *)(m + pg);
  c->magic = CTX_MAGIC_LIVE;
  c->next = g_registry;
  g_registry = c;
  *out = c;
  return HAMSM_OK;
}

static void free_strata(hamsm_ctx *c) {
  if (c->st) {
    for (int s = 0; s < c->S; s++) {
      free(c->st[s].C);
      free(c->st[s].csq);
      free(c->st[s].counts);
    }
    free(c->st);
    c->st = NULL;
  }
  free(c->remap);
  c->remap = NULL;
}

void hamsm_destroy(hamsm_ctx *c) {
  if (!c || !ctx_known(c) || c->magic != CTX_MAGIC_LIVE) return; /* unknown or already destroyed: nothing to free twice */
  free_strata(c);
  for (int d = 0; d < MAX_NDIM; d++) {
    free(c->bounds[d]);
    c->bounds[d] = NULL;
  }
  free(c->centers);
  free(c->ext_own);
  free(c->pca_S2);
  free(c->pca_mu0);
  free(c->seed_chosen);
  free(c->seed_chosen_off);
  c->seed_chosen = c->seed_chosen_off = NULL;
  c->seed_chosen_n = 0;
  c->centers = NULL;
  c->ext_own = NULL;
  c->pca_S2 = c->pca_mu0 = NULL;
  for (size_t i = 0; i < c->n_blocks; i++) free(c->blocks[i]); /* "device" memory dies with its context, as hipFree'd */
  free(c->blocks);
  c->blocks = NULL;
  c->n_blocks = c->cap_blocks = 0;
  c->have_bounds = c->have_mapper = 0;
  c->magic = CTX_MAGIC_DEAD; /* the struct itself stays on the registry (see above) */
}

const char *hamsm_last_error(hamsm_ctx *c) {
  if (!c) return "null ctx";
  if (!ctx_known(c) || c->magic != CTX_MAGIC_LIVE) return g_stale_msg[0] ? g_stale_msg : "stale or foreign ctx handle";
  return c->err;
}

/* "device" memory = host heap, owned by the ctx that allocated it: hamsm_free of a pointer this ctx does not hold
 * (double free, a block of another context) is an error, never a free() of somebody else's memory */
int hamsm_malloc(hamsm_ctx *c, size_t bytes, void **out) {
  CTX(c);
  if (c->n_blocks == c->cap_blocks) {
    size_t cap = c->cap_blocks ? 2 * c->cap_blocks : 64;
    void **nb = (void **)realloc(c->blocks, cap * sizeof(void *));
    if (!nb) return fail(c, HAMSM_ENOMEM, "block table");
    c->blocks = nb;
    c->cap_blocks = cap;
  }
  void *p = malloc(bytes ? bytes : 1);
  if (!p) return fail(c, HAMSM_ENOMEM, "malloc(%zu)", bytes);
  c->blocks[c->n_blocks++] = p;
  *out = p;
  return HAMSM_OK;
}
int hamsm_free(hamsm_ctx *c, void *p) {
  CTX(c);
  if (!p) return HAMSM_OK;
  for (size_t i = c->n_blocks; i-- > 0;)
    if (c->blocks[i] == p) {
      c->blocks[i] = c->blocks[--c->n_blocks];
      free(p);
      return HAMSM_OK;
    }
  return fail(c, HAMSM_EINVAL, "hamsm_free: %p was not returned by hamsm_malloc of this context (double free?)", p);
}
int hamsm_malloc_host(hamsm_ctx *c, size_t bytes, void **out) { return hamsm_malloc(c, bytes, out); }
int hamsm_free_host(hamsm_ctx *c, void *p) { return hamsm_free(c, p); }
/* (ABI 9) process-level host blocks: plain heap here (there is no device to copy from) */
int hamsm_host_block_alloc(size_t bytes, void **out) {
  if (!out) return HAMSM_EINVAL;
  *out = malloc(bytes ? bytes : 1);
  return *out ? HAMSM_OK : HAMSM_ENOMEM;
}
int hamsm_host_block_free(void *p) {
  free(p);
  return HAMSM_OK;
}
int hamsm_memcpy_h2d_async(hamsm_ctx *c, void *d, const void *h, size_t n) {
  CTX(c);
  memcpy(d, h, n);
  return HAMSM_OK;
}
int hamsm_copy_sync(hamsm_ctx *c) {
  CTX(c);
  return HAMSM_OK;
}
/* float32 -> float64 is exact: the staged frames of the reference are float32 on disk, float64 in memory (_data.py:265,565) */
int hamsm_widen_f32(hamsm_ctx *c, const float *src, double *dst, int64_t count) {
  CTX(c);
  if (count < 0 || (count > 0 && (!src || !dst))) return fail(c, HAMSM_EINVAL, "widen_f32: null buffer or negative count");
  if (count > 0 && (const char *)src < (const char *)dst + 8 * count && (const char *)dst < (const char *)src + 4 * count)
    return fail(c, HAMSM_EINVAL, "widen_f32: source and destination overlap");
  for (int64_t i = 0; i < count; i++) dst[i] = (double)src[i];
  return HAMSM_OK;
}
/* (ABI 8) staging's host-side movers, restated without threads: the bytes are what matters, `threads` is advice */
int hamsm_host_copy_parts(hamsm_ctx *c, int64_t n_parts, const void *const *h_src, void *const *h_dst, const int64_t *bytes, int threads) {
  CTX(c);
  (void)threads;
  if (n_parts < 0 || (n_parts > 0 && (!h_src || !h_dst || !bytes))) return fail(c, HAMSM_EINVAL, "host_copy_parts: null table");
  for (int64_t i = 0; i < n_parts; i++) {
    if (bytes[i] < 0) return fail(c, HAMSM_EINVAL, "host_copy_parts: part %lld has a negative size", (long long)i);
    if (bytes[i] > 0 && (!h_src[i] || !h_dst[i])) return fail(c, HAMSM_EINVAL, "host_copy_parts: part %lld has a null pointer", (long long)i);
  }
  for (int64_t i = 0; i < n_parts; i++) memcpy(h_dst[i], h_src[i], (size_t)bytes[i]);
  return HAMSM_OK;
}
int hamsm_host_read_parts(hamsm_ctx *c, int64_t n_parts, const int32_t *fd, const int64_t *file_off, void *const *h_dst,
                          const int64_t *bytes, int threads) {
  CTX(c);
  (void)threads;
  if (n_parts < 0 || (n_parts > 0 && (!fd || !file_off || !h_dst || !bytes))) return fail(c, HAMSM_EINVAL, "host_read_parts: null table");
  for (int64_t i = 0; i < n_parts; i++) {
    if (bytes[i] < 0) return fail(c, HAMSM_EINVAL, "host_read_parts: part %lld has a negative size", (long long)i);
    if (bytes[i] > 0 && (!h_dst[i] || fd[i] < 0 || file_off[i] < 0))
      return fail(c, HAMSM_EINVAL, "host_read_parts: part %lld has a null pointer, a closed descriptor or a negative offset", (long long)i);
  }
  for (int64_t i = 0; i < n_parts; i++) {
    int64_t done = 0;
    while (done < bytes[i]) {
      ssize_t r = pread(fd[i], (char *)h_dst[i] + done, (size_t)(bytes[i] - done), (off_t)(file_off[i] + done));
      if (r < 0 && errno == EINTR) continue;
      if (r < 0) return fail(c, HAMSM_EINVAL, "host_read_parts: part %lld of %lld: %s", (long long)i, (long long)n_parts, strerror(errno));
      if (r == 0) return fail(c, HAMSM_EINVAL, "host_read_parts: part %lld of %lld: short read (the file ends inside the range)", (long long)i, (long long)n_parts);
      done += r;
    }
  }
  return HAMSM_OK;
}
int hamsm_nan_rows(hamsm_ctx *c, const void *X, int dt, int64_t ldx, int D, int64_t n, uint8_t *flags) {
  CTX(c);
  if (dt != HAMSM_F32 && dt != HAMSM_F64) return fail(c, HAMSM_EINVAL, "x_dtype");
  if (n < 0 || D < 1 || ldx < D || (n > 0 && (!X || !flags))) return fail(c, HAMSM_EINVAL, "nan_rows: bad arguments");
  for (int64_t i = 0; i < n; i++) { /* `_data.py:303-313`: a segment whose coordinates contain a NaN */
    int bad = 0;
    for (int k = 0; k < D; k++) {
      const double v = dt == HAMSM_F32 ? (double)((const float *)X)[i * ldx + k] : ((const double *)X)[i * ldx + k];
      bad |= v != v;
    }
    flags[i] = (uint8_t)bad;
  }
  return HAMSM_OK;
}
int hamsm_memcpy_h2d(hamsm_ctx *c, void *d, const void *h, size_t n) {
  CTX(c);
  (void)c;
  memcpy(d, h, n);
  return HAMSM_OK;
}
int hamsm_memcpy_d2h(hamsm_ctx *c, void *h, const void *d, size_t n) {
  CTX(c);
  (void)c;
  memcpy(h, d, n);
  return HAMSM_OK;
}
int hamsm_memset(hamsm_ctx *c, void *d, int b, size_t n) {
  CTX(c);
  (void)c;
  memset(d, b, n);
  return HAMSM_OK;
}
int hamsm_sync(hamsm_ctx *c) {
  CTX(c);
  (void)c;
  return HAMSM_OK;
}

int hamsm_set_geometry(hamsm_ctx *c, int D, int ndim, int S) {
  CTX(c);
  if (D < 1 || ndim < 1 || ndim > MAX_NDIM || S < 1) return fail(c, HAMSM_EINVAL, "bad geometry");
  free_strata(c);
  c->D = D;
  c->ndim = ndim;
  c->S = S;
  c->st = (stratum_t *)calloc((size_t)S, sizeof(stratum_t));
  c->remap = (int32_t *)malloc(sizeof(int32_t) * (size_t)S);
  for (int s = 0; s < S; s++) c->remap[s] = s; /* stratified_clustering.py:85 */
  return HAMSM_OK;
}

int hamsm_set_mapper_rectilinear(hamsm_ctx *c, const int32_t *nb, const float *b) {
  CTX(c);
  if (!c->st) return fail(c, HAMSM_EINVAL, "set_geometry first");
  long nbins = 1;
  for (int d = 0; d < c->ndim; d++) {
    if (nb[d] < 2) return fail(c, HAMSM_EINVAL, "need >=2 boundaries per dim");
    free(c->bounds[d]);
    c->bounds[d] = (float *)malloc(sizeof(float) * (size_t)nb[d]);
    memcpy(c->bounds[d], b, sizeof(float) * (size_t)nb[d]);
    c->nb[d] = nb[d];
    b += nb[d];
    nbins *= nb[d] - 1;
  }
  if (nbins != c->S) return fail(c, HAMSM_EINVAL, "mapper has %ld bins, geometry says %d", nbins, c->S);
  c->mapper_kind = 0;
  c->ext_set = 0;
  c->have_mapper = 1;
  return HAMSM_OK;
}

int hamsm_set_mapper_voronoi(hamsm_ctx *c, int n_centers, const float *centers) {
  CTX(c);
  if (!c->st) return fail(c, HAMSM_EINVAL, "set_geometry first");
  if (n_centers != c->S) return fail(c, HAMSM_EINVAL, "mapper has %d centres, geometry says %d bins", n_centers, c->S);
  free(c->centers);
  c->centers = (float *)malloc(sizeof(float) * (size_t)n_centers * c->ndim);
  memcpy(c->centers, centers, sizeof(float) * (size_t)n_centers * c->ndim);
  c->mapper_kind = 1;
  c->ext_set = 0;
  c->have_mapper = 1;
  return HAMSM_OK;
}

int hamsm_set_mapper_external(hamsm_ctx *c) {
  CTX(c);
  if (!c->st) return fail(c, HAMSM_EINVAL, "set_geometry first");
  c->mapper_kind = 2;
  c->ext_bins = NULL;
  c->ext_set = 0;
  c->have_mapper = 1;
  return HAMSM_OK;
}

int hamsm_set_frame_bins(hamsm_ctx *c, const int32_t *bins) {
  CTX(c);
  if (c->mapper_kind != 2) return fail(c, HAMSM_EINVAL, "frame bins are for the external mapper only");
  c->ext_bins = bins;
  c->ext_set = bins != NULL;
  return HAMSM_OK;
}

int hamsm_set_frame_bins_host(hamsm_ctx *c, const int32_t *bins, int64_t n) {
  CTX(c);
  if (c->mapper_kind != 2) return fail(c, HAMSM_EINVAL, "frame bins are for the external mapper only");
  if (n <= 0 || !bins) return fail(c, HAMSM_EINVAL, "frame bins missing");
  free(c->ext_own);
  c->ext_own = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  memcpy(c->ext_own, bins, sizeof(int32_t) * (size_t)n);
  c->ext_bins = c->ext_own;
  c->ext_set = 1;
  return HAMSM_OK;
}

int hamsm_set_remap(hamsm_ctx *c, const int32_t *r) {
  CTX(c);
  if (!c->st) return fail(c, HAMSM_EINVAL, "set_geometry first");
  for (int s = 0; s < c->S; s++)
    if (r[s] < 0 || r[s] >= c->S) return fail(c, HAMSM_EINVAL, "remap[%d]=%d out of range", s, r[s]);
  memcpy(c->remap, r, sizeof(int32_t) * (size_t)c->S);
  return HAMSM_OK;
}

int hamsm_set_bounds(hamsm_ctx *c, const double *basis, const double *target) {
  CTX(c);
  if (!c->st) return fail(c, HAMSM_EINVAL, "set_geometry first");
  for (int d = 0; d < c->ndim; d++) {
    c->basis[d][0] = basis[2 * d];
    c->basis[d][1] = basis[2 * d + 1];
    c->target[d][0] = target[2 * d];
    c->target[d][1] = target[2 * d + 1];
  }
  c->have_bounds = 1;
  return HAMSM_OK;
}

/* ||c||^2: fma chain over ascending feature index (sklearn uses einsum, whose SIMD order is
 * machine dependent; callers that need sklearn's bits pass h_csq, see hamsm.h). */
static double sqnorm_chain(const double *v, int D) {
  double a = 0.0;
  for (int k = 0; k < D; k++) a = __builtin_fma(v[k], v[k], a);
  return a;
}

int hamsm_set_centroids(hamsm_ctx *c, int s, int K, const double *C, const double *csq, const double *counts) {
  CTX(c);
  if (!c->st || s < 0 || s >= c->S || K < 0) return fail(c, HAMSM_EINVAL, "bad stratum/K");
  stratum_t *t = &c->st[s];
  free(t->C);
  free(t->csq);
  free(t->counts);
  t->C = t->csq = t->counts = NULL;
  t->K = K;
  if (K == 0) return HAMSM_OK;
  t->C = (double *)malloc(sizeof(double) * (size_t)K * c->D);
  t->csq = (double *)malloc(sizeof(double) * (size_t)K);
  t->counts = (double *)calloc((size_t)K, sizeof(double));
  memcpy(t->C, C, sizeof(double) * (size_t)K * c->D);
  for (int j = 0; j < K; j++) t->csq[j] = csq ? csq[j] : sqnorm_chain(t->C + (size_t)j * c->D, c->D);
  if (counts) memcpy(t->counts, counts, sizeof(double) * (size_t)K);
  return HAMSM_OK;
}

int hamsm_get_centroids(hamsm_ctx *c, int s, double *C, double *counts, double *csq) {
  CTX(c);
  if (!c->st || s < 0 || s >= c->S) return fail(c, HAMSM_EINVAL, "bad stratum");
  stratum_t *t = &c->st[s];
  if (C) memcpy(C, t->C, sizeof(double) * (size_t)t->K * c->D);
  if (counts) memcpy(counts, t->counts, sizeof(double) * (size_t)t->K);
  if (csq) memcpy(csq, t->csq, sizeof(double) * (size_t)t->K);
  return HAMSM_OK;
}

int hamsm_num_centroids(hamsm_ctx *c, int s) {
  CTX(c);
  if (!c->st || s < 0 || s >= c->S) return fail(c, HAMSM_EINVAL, "bad stratum");
  return c->st[s].K;
}

/* --- a3: westpa RectilinearBinMapper.assign restated (third-party, source absent):
 * float32 compare, half-open bins, row-major over dims.  -1 = outside the bin space. */
static int bin_of(const hamsm_ctx *c, const double *p, int64_t i) {
  if (c->mapper_kind == 2) return c->ext_bins[i]; /* the mapper's own assign ran on the host */
  if (c->mapper_kind == 1) { /* VoronoiBinMapper, Euclidean dfunc: float32 coordinates, float64 squared distance, first minimum */
    int best = -1;
    double bd = 0.0;
    for (int k = 0; k < c->S; k++) {
      double d2 = 0.0;
      for (int d = 0; d < c->ndim; d++) {
        const double t = (double)(float)p[d] - (double)c->centers[k * c->ndim + d];
        d2 = d2 + t * t;
      }
      if (best < 0 || d2 < bd) {
        bd = d2;
        best = k;
      }
    }
    return bd == bd ? best : -1;
  }
  int idx = 0;
  for (int d = 0; d < c->ndim; d++) {
    float pf = (float)p[d];
    const float *b = c->bounds[d];
    int n = c->nb[d], lo = 0, hi = n; /* first index with b[i] > pf (searchsorted side='right') */
    while (lo < hi) {
      int mid = (lo + hi) / 2;
      if (b[mid] <= pf) lo = mid + 1; /* NaN: comparison false -> hi shrinks -> lo=0 -> j=-1 */
      else hi = mid;
    }
    int j = lo - 1;
    if (j < 0 || j >= n - 1) return -1;
    idx = idx * (n - 1) + j;
  }
  return idx;
}

static int take_frame_bins(hamsm_ctx *c, const char *what) {
  if (c->mapper_kind != 2) return HAMSM_OK;
  if (!c->ext_set) return fail(c, HAMSM_EINVAL, "%s: external mapper -- call hamsm_set_frame_bins first", what);
  c->ext_set = 0;
  return HAMSM_OK;
}

/* --- a4: msm_we.py:480-492, 515-527: all dims strictly inside (lo, hi), float64 */
static int in_box(const double (*box)[2], int ndim, const double *p) {
  for (int d = 0; d < ndim; d++)
    if (!(p[d] > box[d][0] && p[d] < box[d][1])) return 0;
  return 1;
}

int hamsm_classify(hamsm_ctx *c, const double *p, int64_t n, int apply_remap, int32_t *bin, uint8_t *flags) {
  CTX(c);
  if (!c->have_mapper || !c->have_bounds) return fail(c, HAMSM_EINVAL, "mapper/bounds not set");
  if (n <= 0) return HAMSM_OK;
  if (take_frame_bins(c, "classify")) return HAMSM_EINVAL;
  for (int64_t i = 0; i < n; i++) {
    const double *pi = p + i * c->ndim;
    int b = bin_of(c, pi, i);
    if (b >= 0 && apply_remap) b = c->remap[b];
    if (bin) bin[i] = b;
    if (flags) flags[i] = (uint8_t)((in_box(c->basis, c->ndim, pi) ? 1 : 0) | (in_box(c->target, c->ndim, pi) ? 2 : 0));
  }
  return HAMSM_OK;
}

static inline double xval(const void *X, int dt, int64_t i) {
  return dt == HAMSM_F32 ? (double)((const float *)X)[i] : ((const double *)X)[i];
}

/* --- sklearn _k_means_lloyd.pyx:188-218 (_update_chunk_dense, update_centers=False):
 * pairwise = ||c||^2 - 2 x.c via dgemm (fma chain per element), first strict minimum. */
static int argmin_row(const double *x, const stratum_t *t, int D) {
  int best = 0;
  double bd = 0.0;
  for (int j = 0; j < t->K; j++) {
    const double *cj = t->C + (size_t)j * D;
    double acc = 0.0;
    for (int k = 0; k < D; k++) acc = __builtin_fma(x[k], cj[k], acc);
    double d = __builtin_fma(-2.0, acc, t->csq[j]);
    if (j == 0 || d < bd) {
      bd = d;
      best = j;
    }
  }
  return best;
}

/* --- a9: StratifiedClusters.predict, stratified_clustering.py:129-212 */
int hamsm_assign(hamsm_ctx *c, const void *X, int dt, int64_t ldx, const double *p, int64_t n, int32_t *labels,
                 int32_t *seen) {
  CTX(c);
  if (!c->have_mapper || !c->have_bounds) return fail(c, HAMSM_EINVAL, "mapper/bounds not set");
  if (dt != HAMSM_F32 && dt != HAMSM_F64) return fail(c, HAMSM_EINVAL, "x_dtype");
  if (ldx < c->D) return fail(c, HAMSM_EINVAL, "ldx < D");
  if (n <= 0) return HAMSM_OK;
  if (take_frame_bins(c, "assign")) return HAMSM_EINVAL;
  int D = c->D;
  /* :143-150 total_clusters over all strata; :178-185 offset of earlier strata */
  int Kfit = 0;
  int *off = (int *)malloc(sizeof(int) * (size_t)c->S);
  for (int s = 0; s < c->S; s++) {
    off[s] = Kfit;
    Kfit += c->st[s].K;
  }
  double *x = (double *)malloc(sizeof(double) * (size_t)D);
  int rc = HAMSM_OK;
  for (int64_t i = 0; i < n && rc == HAMSM_OK; i++) {
    const double *pi = p + i * c->ndim;
    int b = bin_of(c, pi, i); /* :134 assign is evaluated for every frame, basis/target included */
    if (b < 0 || b >= c->S) {
      rc = fail(c, HAMSM_ERANGE, "frame %lld: pcoord outside the bin space", (long long)i);
      break;
    }
    int s = c->remap[b]; /* :135 */
    if (in_box(c->target, c->ndim, pi)) { /* :159 target first */
      labels[i] = Kfit + 1;
      if (seen) seen[s] = 1;
    } else if (in_box(c->basis, c->ndim, pi)) { /* :165 */
      labels[i] = Kfit;
      if (seen) seen[c->S + s] = 1;
    } else {
      const stratum_t *t = &c->st[s];
      if (t->K == 0) { /* :187-189 */
        rc = fail(c, HAMSM_EUNFIT, "Not initialized in seg %lld, bin %d", (long long)i, s);
        break;
      }
      for (int k = 0; k < D; k++) x[k] = xval(X, dt, i * ldx + k);
      labels[i] = off[s] + argmin_row(x, t, D); /* :195 */
    }
  }
  free(x);
  free(off);
  return rc;
}

int hamsm_labels_to_host(hamsm_ctx *c, const int32_t *pa, const int32_t *ch, int64_t n, int64_t *h_child, int64_t *h_pairs) {
  CTX(c);
  for (int64_t i = 0; i < n; i++) {
    if (h_child) h_child[i] = ch[i];
    if (h_pairs) {
      h_pairs[2 * i] = pa[i];
      h_pairs[2 * i + 1] = ch[i];
    }
  }
  return HAMSM_OK;
}

/* (ABI 10) both sides of the same segments: the parent frames, then the child frames (_clustering.py:1299-1301, :1315-1316) */
int hamsm_assign_pair(hamsm_ctx *c, const void *Xp, const void *Xc, int dt, int64_t ldx, const double *p0, const double *p1,
                      int64_t n, int32_t *parent, int32_t *child, int32_t *seen) {
  CTX(c);
  if (c->have_mapper && c->mapper_kind == 2 && n > 0)
    return fail(c, HAMSM_EINVAL, "assign_pair: an external mapper's frame bins are set per side (two hamsm_assign calls)");
  int rc = hamsm_assign(c, Xp, dt, ldx, p0, n, parent, seen);
  if (rc) return rc;
  return hamsm_assign(c, Xc, dt, ldx, p1, n, child, seen);
}

int hamsm_predict_host(hamsm_ctx *c, const void *X, int dt, const double *p, int64_t n, int32_t *labels,
                       int32_t *seen) {
  CTX(c);
  return hamsm_assign(c, X, dt, c->D, p, n, labels, seen);
}

/* --- a10: _fluxmatrix.py:97-164 (build_flux_matrix) + :325-327,342 (sum of per-iteration dense
 * matrices in ascending iteration order).  Per iteration the COO->dense conversion adds duplicates
 * in segment order (scipy coo_todense); iterations that do not touch a cell add +0.0, which is
 * exact, so a lazily flushed per-cell partial is bit-identical to the dense sums. */
int hamsm_flux_accumulate(hamsm_ctx *c, const int32_t *pa, const int32_t *ch, const double *p0, const double *p1,
                          const double *w, int64_t n, int K, double *F, int mode, const int64_t *offs,
                          int64_t n_iters) {
  CTX(c);
  (void)mode;
  if (!c->have_bounds) return fail(c, HAMSM_EINVAL, "bounds not set");
  int64_t N2 = (int64_t)(K + 2) * (K + 2);
  double *part = (double *)calloc((size_t)N2, sizeof(double));
  int64_t *last = (int64_t *)malloc(sizeof(int64_t) * (size_t)N2);
  int64_t *touched = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  if (!part || !last || !touched) return fail(c, HAMSM_ENOMEM, "flux scratch");
  for (int64_t i = 0; i < N2; i++) last[i] = -1;
  int64_t one[2] = {0, n};
  if (!offs) {
    offs = one;
    n_iters = 1;
  }
  int rc = HAMSM_OK;
  for (int64_t t = 0; t < n_iters && rc == HAMSM_OK; t++) {
    int64_t nt = 0;
    for (int64_t i = offs[t]; i < offs[t + 1]; i++) {
      int a = pa[i], b = ch[i];
      if (in_box(c->target, c->ndim, p1 + i * c->ndim)) b = K + 1; /* :135 */
      if (in_box(c->basis, c->ndim, p0 + i * c->ndim)) a = K;      /* :136 */
      if (in_box(c->basis, c->ndim, p1 + i * c->ndim)) b = K;      /* :137 */
      if (a < 0 || b < 0 || a >= K + 2 || b >= K + 2) {
        rc = fail(c, HAMSM_ERANGE, "segment %lld: transition %d -> %d outside %d states", (long long)i, a, b, K + 2);
        break;
      }
      int64_t cell = (int64_t)a * (K + 2) + b;
      if (last[cell] != t) {
        last[cell] = t;
        part[cell] = 0.0;
        touched[nt++] = cell;
      }
      part[cell] = part[cell] + w[i];
    }
    for (int64_t k = 0; k < nt; k++) F[touched[k]] = F[touched[k]] + part[touched[k]];
  }
  free(part);
  free(last);
  free(touched);
  return rc;
}

int hamsm_scale(hamsm_ctx *c, double *F, int64_t count, double div) {
  CTX(c);
  (void)c;
  for (int64_t i = 0; i < count; i++) F[i] = F[i] / div; /* _fluxmatrix.py:342 */
  return HAMSM_OK;
}

/* --- a11: _clustering.py:1559-1591.  (The reference averages with np.nanmean = pairwise sum;
 * the mean only feeds an argsort, so the contract is sum in ascending frame order, |rel| 1e-12.) */
int hamsm_cluster_stats(hamsm_ctx *c, const int32_t *lab, const double *p, int64_t n, int nstates, double *sum,
                        double *mn, double *mx, int64_t *cnt) {
  CTX(c);
  for (int64_t i = 0; i < n; i++) {
    int l = lab[i];
    if (l < 0 || l >= nstates) return fail(c, HAMSM_ERANGE, "label %d outside %d states", l, nstates);
    double v = p[i * c->ndim];
    sum[l] = sum[l] + v;
    if (v < mn[l]) mn[l] = v;
    if (v > mx[l]) mx[l] = v;
    cnt[l]++;
  }
  return HAMSM_OK;
}

/* numpy pairwise summation (numpy/core/src/umath/loops_utils.h.src, pairwise_sum): what
 * np.sum(fluxmatrix, 1) computes for a contiguous row -- reproduced so that Tmatrix is bit-equal
 * to AnalysisMixin.get_Tmatrix (_analysis.py:48). */
static double np_pairwise(const double *a, int64_t n) {
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

/* --- a13 feed: _analysis.py:44-79 */
int hamsm_tmatrix_csr(hamsm_ctx *c, const double *F, int ns, const int32_t *basis, int nb, const int32_t *target,
                      int nt, int64_t *indptr, int32_t *indices, double *data, int64_t cap, int64_t *nnz_out) {
  CTX(c);
  int64_t nnz = 0;
  indptr[0] = 0;
  for (int i = 0; i < ns; i++) {
    int is_t = 0;
    for (int k = 0; k < nt; k++) is_t |= (target[k] == i);
    const double *row = F + (size_t)i * ns;
    double out = np_pairwise(row, ns);
    for (int j = 0; j < ns; j++) {
      double v;
      if (is_t) { /* :74-77 sink rows recycle uniformly into the basis */
        v = 0.0;
        for (int k = 0; k < nb; k++)
          if (basis[k] == j) v = 1.0 / (double)nb;
      } else {
        v = row[j];
        if (out > 0) v = v / out;            /* :53-56 */
        if (out == 0.0 && j == i) v = 1.0;   /* :60-61 */
      }
      if (v != 0.0) {
        if (indices || data) { /* (both NULL: count only) */
          if (nnz >= cap) return fail(c, HAMSM_ENOMEM, "csr capacity %lld too small", (long long)cap);
          indices[nnz] = j;
          data[nnz] = v;
        }
        nnz++;
      }
    }
    indptr[i + 1] = nnz;
  }
  *nnz_out = nnz;
  return HAMSM_OK;
}

/* --- a12 on the device: see include/hamsm.h */
int hamsm_pattern_csr(hamsm_ctx *c, const double *F, int ns, int64_t *indptr, int32_t *indices, int64_t cap, int64_t *nnz_out) {
  CTX(c);
  if (ns <= 0 || !indptr || !nnz_out) return fail(c, HAMSM_EINVAL, "pattern_csr: arguments");
  int64_t nnz = 0;
  indptr[0] = 0;
  for (int i = 0; i < ns; i++) {
    for (int j = 0; j < ns; j++)
      if (F[(size_t)i * ns + j] != 0.0) {
        if (indices) {
          if (nnz >= cap) return fail(c, HAMSM_ENOMEM, "csr capacity %lld too small", (long long)cap);
          indices[nnz] = j;
        }
        nnz++;
      }
    indptr[i + 1] = nnz;
  }
  *nnz_out = nnz;
  return HAMSM_OK;
}

/* numpy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum) for a contiguous run */
int hamsm_np_sum(hamsm_ctx *c, const double *a, int64_t n, double *out) {
  CTX(c);
  if (!out) return fail(c, HAMSM_EINVAL, "np_sum: out");
  double t = 0.0; /* numpy's reduction loop: chunks of the ufunc buffer size (8192), each pairwise, added in order */
  for (int64_t off = 0; off < n; off += 8192) t = t + np_pairwise(a + off, n - off < 8192 ? n - off : 8192);
  *out = t;
  return HAMSM_OK;
}

int hamsm_permute_scale(hamsm_ctx *c, const double *F, int ns, const int32_t *perm, double *out, double div, int transpose) {
  CTX(c);
  if (ns <= 0 || !perm || F == out) return fail(c, HAMSM_EINVAL, "permute_scale: arguments");
  for (int i = 0; i < ns; i++)
    if (perm[i] < 0 || perm[i] >= ns) return fail(c, HAMSM_EINVAL, "permute_scale: perm[%d] = %d", i, perm[i]);
  for (int i = 0; i < ns; i++)
    for (int j = 0; j < ns; j++) {
      const double v = transpose ? F[(size_t)perm[j] * ns + perm[i]] : F[(size_t)perm[i] * ns + perm[j]];
      out[(size_t)i * ns + j] = div == 1.0 ? v : v / div;
    }
  return HAMSM_OK;
}

/* --- a2: sklearn _BasePCA._transform (X @ components_.T, then -= mean_ @ components_.T, then /= scale) with every
 *     dot product an fma chain over ascending k */
int hamsm_project(hamsm_ctx *c, const void *X, int dt, int64_t ldx, int64_t n, int Draw, const double *mean, const double *W,
                  int d, const double *scale, double *Y) {
  CTX(c);
  if (Draw < 1 || d < 1 || !W || ldx < Draw) return fail(c, HAMSM_EINVAL, "bad projection shape");
  if (dt != HAMSM_F32 && dt != HAMSM_F64) return fail(c, HAMSM_EINVAL, "bad x_dtype");
  double *m = (double *)calloc((size_t)d, sizeof(double));
  if (mean)
    for (int j = 0; j < d; j++) {
      double a = 0.0;
      for (int k = 0; k < Draw; k++) a = __builtin_fma(W[(size_t)j * Draw + k], mean[k], a);
      m[j] = a;
    }
  for (int64_t i = 0; i < n; i++)
    for (int j = 0; j < d; j++) {
      double a = 0.0;
      for (int k = 0; k < Draw; k++) a = __builtin_fma(W[(size_t)j * Draw + k], xval(X, dt, i * ldx + k), a);
      if (mean) a = a - m[j];
      if (scale) a = a / scale[j];
      Y[i * d + j] = a;
    }
  free(m);
  return HAMSM_OK;
}

/* --- f3: streaming PCA fit by covariance accumulation (opt-in replacement of the IncrementalPCA loop of
 *     _dimensionality.py:69-108, 171-241; include/hamsm.h states the order).  The sums of the shifted features
 *     z = x - (first row), with a ones column carrying S1: per call G strips of L rows, per strip and pair a <= b one fma
 *     chain over the rows, strips added in order inside groups of 64, groups added in order, the call's total added to the
 *     running sum. */
int hamsm_pca_begin(hamsm_ctx *c, int Draw) {
  CTX(c);
  if (Draw < 1 || Draw > 16384) return fail(c, HAMSM_EINVAL, "pca_begin: %d features", Draw);
  free(c->pca_S2);
  free(c->pca_mu0);
  const int Dp = (Draw + 1 + 31) & ~31;
  c->pca_S2 = (double *)calloc((size_t)Dp * Dp, sizeof(double));
  c->pca_mu0 = (double *)calloc((size_t)Dp, sizeof(double));
  if (!c->pca_S2 || !c->pca_mu0) return fail(c, HAMSM_ENOMEM, "pca sums");
  c->pca_D = Draw;
  c->pca_n = 0;
  return HAMSM_OK;
}

int hamsm_pca_accumulate(hamsm_ctx *c, const void *X, int dt, int64_t ldx, int64_t n) {
  CTX(c);
  if (!c->pca_S2) return fail(c, HAMSM_EINVAL, "pca_accumulate before pca_begin");
  if (dt != HAMSM_F32 && dt != HAMSM_F64) return fail(c, HAMSM_EINVAL, "bad x_dtype");
  const int D = c->pca_D, Dp = (D + 1 + 31) & ~31;
  if (ldx < D) return fail(c, HAMSM_EINVAL, "ldx < D");
  if (n <= 0) return HAMSM_OK;
  if (c->pca_n == 0)
    for (int a = 0; a < D; a++) c->pca_mu0[a] = xval(X, dt, a);
  int64_t G = (n + 511) / 512;
  const int64_t cap = ((int64_t)1 << 25) / ((int64_t)Dp * Dp);
  if (G > 8192) G = 8192;
  if (G > cap) G = cap;
  if (G < 1) G = 1;
  int64_t L = (n + G - 1) / G;
  L = (L + 3) & ~(int64_t)3;
  double *z = (double *)malloc(sizeof(double) * (size_t)L * (D + 1)); /* a strip's shifted rows, feature-major */
  double *T = (double *)malloc(sizeof(double) * (size_t)(D + 1) * (D + 1)); /* the call's total */
  double *Q = (double *)malloc(sizeof(double) * (size_t)(D + 1) * (D + 1)); /* the current group of 64 strips */
  if (!z || !T || !Q) {
    free(z);
    free(T);
    free(Q);
    return fail(c, HAMSM_ENOMEM, "pca strip");
  }
  for (int64_t g = 0; g < G; g++) {
    const int64_t f0 = g * L, fend = f0 + L < n ? f0 + L : n, m = fend > f0 ? fend - f0 : 0;
    for (int a = 0; a <= D; a++)
      for (int64_t f = 0; f < m; f++) z[(size_t)a * L + f] = a < D ? xval(X, dt, (f0 + f) * ldx + a) - c->pca_mu0[a] : 1.0;
    for (int a = 0; a <= D; a++)
      for (int b = a; b <= D; b++) {
        double acc = 0.0;
        const double *za = z + (size_t)a * L, *zb = z + (size_t)b * L;
        for (int64_t f = 0; f < m; f++) acc = __builtin_fma(za[f], zb[f], acc);
        double *q = &Q[(size_t)a * (D + 1) + b];
        *q = g % 64 == 0 ? acc : *q + acc;
      }
    if (g % 64 == 63 || g == G - 1) /* the group is complete: its sum joins the call's total, groups in order */
      for (int a = 0; a <= D; a++)
        for (int b = a; b <= D; b++) {
          double *t = &T[(size_t)a * (D + 1) + b];
          *t = g < 64 ? Q[(size_t)a * (D + 1) + b] : *t + Q[(size_t)a * (D + 1) + b];
        }
  }
  for (int a = 0; a <= D; a++)
    for (int b = a; b <= D; b++) c->pca_S2[(size_t)a * Dp + b] = c->pca_S2[(size_t)a * Dp + b] + T[(size_t)a * (D + 1) + b];
  free(z);
  free(T);
  free(Q);
  c->pca_n += n;
  return HAMSM_OK;
}

int hamsm_pca_finish(hamsm_ctx *c, int64_t *n_total, double *h_shift, double *h_S1, double *h_S2) {
  CTX(c);
  if (!c->pca_S2) return fail(c, HAMSM_EINVAL, "pca_finish before pca_begin");
  const int D = c->pca_D, Dp = (D + 1 + 31) & ~31;
  if (n_total) *n_total = c->pca_n;
  for (int a = 0; a < D; a++) {
    if (h_shift) h_shift[a] = c->pca_mu0[a];
    if (h_S1) h_S1[a] = c->pca_S2[(size_t)a * Dp + D];
    if (h_S2)
      for (int b = 0; b < D; b++) h_S2[(size_t)a * D + b] = a <= b ? c->pca_S2[(size_t)a * Dp + b] : c->pca_S2[(size_t)b * Dp + a];
  }
  free(c->pca_S2);
  free(c->pca_mu0);
  c->pca_S2 = c->pca_mu0 = NULL;
  c->pca_D = 0;
  c->pca_n = 0;
  return HAMSM_OK;
}

/* --- a1: tests/test_msm_we.py:19-36 -> MDAnalysis distances.dist(...)[2] = lib.distances.calc_bonds without a box
 *     (float32 positions, float32 difference, float64 squares / sum / sqrt).  Restated from the published algorithm:
 *     MDAnalysis is not available here, so this function is NOT pinned by reference output ("parity unpinned"). */
int hamsm_featurize_dist(hamsm_ctx *c, const void *coords, int dt, int64_t ldc, int64_t n, int n_atoms, const int32_t *idx,
                         int n_sel, const float *ref, double *X, int64_t ldx) {
  CTX(c);
  if (dt != HAMSM_F32 && dt != HAMSM_F64) return fail(c, HAMSM_EINVAL, "bad c_dtype");
  if (n_atoms < 1 || n_sel < 1 || !idx || !ref || ldc < 3 * (int64_t)n_atoms || ldx < n_sel)
    return fail(c, HAMSM_EINVAL, "bad featurizer shape");
  for (int a = 0; a < n_sel; a++)
    if (idx[a] < 0 || idx[a] >= n_atoms) return fail(c, HAMSM_EINVAL, "atom index %d out of range", (int)idx[a]);
  for (int64_t i = 0; i < n; i++)
    for (int a = 0; a < n_sel; a++) {
      double d[3];
      for (int q = 0; q < 3; q++) {
        float x = (float)xval(coords, dt, i * ldc + 3 * (int64_t)idx[a] + q);
        float df = x - ref[3 * a + q];
        d[q] = (double)df;
      }
      double rsq = d[0] * d[0];
      rsq = rsq + d[1] * d[1];
      rsq = rsq + d[2] * d[2];
      X[i * ldx + a] = sqrt(rsq);
    }
  return HAMSM_OK;
}

/* --- f4: the grouping of update_cluster_structures (_clustering.py:1460-1500): segments appended to their cluster's list in
 * segment order = a stable counting sort of the segment indices by label */
int hamsm_group_by_label(hamsm_ctx *c, const int32_t *lab, int64_t n, int ns, int32_t *order, int64_t *indptr) {
  CTX(c);
  if (ns <= 0 || !indptr) return fail(c, HAMSM_EINVAL, "group_by_label: nstates / indptr");
  for (int s = 0; s <= ns; s++) indptr[s] = 0;
  for (int64_t i = 0; i < n; i++) {
    if (lab[i] < 0 || lab[i] >= ns) return fail(c, HAMSM_ERANGE, "group_by_label: index out of range at element %lld (value %d)", (long long)i, lab[i]);
    indptr[lab[i] + 1]++;
  }
  for (int s = 0; s < ns; s++) indptr[s + 1] += indptr[s];
  int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)ns);
  for (int s = 0; s < ns; s++) fill[s] = indptr[s];
  for (int64_t i = 0; i < n; i++) order[fill[lab[i]]++] = (int32_t)i;
  free(fill);
  return HAMSM_OK;
}

/* --- a5 support: _clustering.py:872-886 */
int hamsm_train_prepass(hamsm_ctx *c, const double *p, int64_t n, const int64_t *offs, int64_t n_iters, int32_t *bin,
                        int64_t *hist) {
  CTX(c);
  if (!c->have_mapper || !c->have_bounds) return fail(c, HAMSM_EINVAL, "mapper/bounds not set");
  if (n_iters <= 0) return HAMSM_OK;
  if (take_frame_bins(c, "train_prepass")) return HAMSM_EINVAL;
  memset(hist, 0, sizeof(int64_t) * (size_t)n_iters * c->S);
  for (int64_t t = 0; t < n_iters; t++)
    for (int64_t i = offs[t]; i < offs[t + 1]; i++) {
      const double *pi = p + i * c->ndim;
      int b;
      if (in_box(c->target, c->ndim, pi) || in_box(c->basis, c->ndim, pi)) b = -1; /* :872-874 */
      else {
        b = bin_of(c, pi, i); /* :877, no we_remap here */
        if (b < 0 || b >= c->S) return fail(c, HAMSM_ERANGE, "frame %lld: pcoord outside the bin space", (long long)i);
        hist[t * c->S + b]++;
      }
      bin[i] = b;
    }
  (void)n;
  return HAMSM_OK;
}

int hamsm_train_build_rows(hamsm_ctx *c, const int32_t *bin, int64_t n, const int32_t *call_remap, int32_t *rows,
                           int64_t *counts, int64_t *starts) {
  CTX(c);
  if (!c->st || c->S < 1) return fail(c, HAMSM_EINVAL, "set_geometry first");
  for (int s = 0; s < c->S; s++) /* the remap row names strata */
    if (call_remap[s] < 0 || call_remap[s] >= c->S) return fail(c, HAMSM_EINVAL, "call_remap[%d] = %d is not a stratum", s, call_remap[s]);
  memset(counts, 0, sizeof(int64_t) * (size_t)c->S);
  for (int64_t i = 0; i < n; i++)
    if (bin[i] >= 0) counts[call_remap[bin[i]]]++;
  int64_t acc = 0;
  for (int s = 0; s < c->S; s++) {
    starts[s] = acc;
    acc += counts[s];
  }
  const size_t nS = c->S > 0 ? (size_t)c->S : 1;
  int64_t *fill = (int64_t *)calloc(nS, sizeof(int64_t));
  int64_t rank = 0; /* position in the basis/target-filtered list (:874); applied to the unfiltered rows (:892-896) */
  for (int64_t i = 0; i < n; i++)
    if (bin[i] >= 0) {
      int s = call_remap[bin[i]];
      rows[starts[s] + fill[s]++] = (int32_t)rank;
      rank++;
    }
  free(fill);
  return HAMSM_OK;
}

/* batch row r -> frame index */
static inline int64_t frame_of(const int32_t *rows, const int64_t *sub, int64_t r) {
  if (sub) r = sub[r];
  return rows ? rows[r] : r;
}

static void load_row(double *dst, const void *X, int dt, int64_t ldx, int64_t f, int D) {
  for (int k = 0; k < D; k++) dst[k] = xval(X, dt, f * ldx + k);
}

/* potentials (sklearn: `@` via BLAS, order unspecified).  Fixed order, chosen so that a 1024-thread
 * workgroup reproduces it without serial bottlenecks: POT_LANES strided partials (lane t sums
 * d[i]*w[i] over i = t, t+POT_LANES, ... ascending, multiply then add), POT_GROUP consecutive lanes
 * are added in lane order, and the group sums are added in group order. */
static double pot_sum(const double *d, const double *w, int64_t n) {
  double grp[POT_LANES / POT_GROUP];
  for (int g = 0; g < POT_LANES / POT_GROUP; g++) {
    double gs = 0.0;
    for (int t = g * POT_GROUP; t < (g + 1) * POT_GROUP; t++) {
      double a = 0.0;
      for (int64_t i = t; i < n; i += POT_LANES) a = a + d[i] * w[i];
      gs = gs + a;
    }
    grp[g] = gs;
  }
  double s = 0.0;
  for (int g = 0; g < POT_LANES / POT_GROUP; g++) s = s + grp[g];
  return s;
}

/* --- a6: sklearn _kmeans.py:214-276 */
int hamsm_kmeans_seed(hamsm_ctx *c, int s, const void *X, int dt, int64_t ldx, const int32_t *rows, int64_t n_b,
                      const double *wf, const int64_t *sub, int64_t n_sub, int K, int64_t first, const double *u,
                      int L, int64_t *chosen) {
  CTX(c);
  if (!c->st || s < 0 || s >= c->S) return fail(c, HAMSM_EINVAL, "bad stratum");
  int D = c->D;
  int64_t n = sub ? n_sub : n_b;
  if (K < 1 || n < 1 || first < 0 || first >= n || L < 1 || L > 64) return fail(c, HAMSM_EINVAL, "bad seed args");
  double *Xb = (double *)malloc(sizeof(double) * (size_t)n * D);
  double *w = (double *)malloc(sizeof(double) * (size_t)n);
  double *xsq = (double *)malloc(sizeof(double) * (size_t)n);
  double *closest = (double *)malloc(sizeof(double) * (size_t)n);
  double *cand = (double *)malloc(sizeof(double) * (size_t)n * L);
  double *cum = (double *)malloc(sizeof(double) * (size_t)n);
  double *C = (double *)malloc(sizeof(double) * (size_t)K * D);
  for (int64_t i = 0; i < n; i++) {
    int64_t f = frame_of(rows, sub, i);
    load_row(Xb + i * D, X, dt, ldx, f, D);
    w[i] = wf ? wf[f] : 1.0;
    xsq[i] = sqnorm_chain(Xb + i * D, D); /* partial_fit: row_norms(X, squared=True), _kmeans.py:2246 */
  }
  /* :225-232 first centre; :235-238 closest = ||x_first - x||^2, pot */
  memcpy(C, Xb + first * D, sizeof(double) * D);
  if (chosen) chosen[0] = first;
#define DIST(ci, i)                                                                  \
  ({                                                                                 \
    double acc_ = 0.0;                                                               \
    const double *a_ = Xb + (ci)*D, *b_ = Xb + (i)*D;                                \
    for (int k_ = 0; k_ < D; k_++) acc_ = __builtin_fma(a_[k_], b_[k_], acc_);      \
    double v_ = ((-2.0 * acc_) + xsq[ci]) + xsq[i]; /* euclidean_distances order */ \
    v_ > 0.0 ? v_ : 0.0;                           /* np.maximum(d, 0); NaN -> 0 is not reproduced: inputs are finite */ \
  })
  for (int64_t i = 0; i < n; i++) closest[i] = DIST(first, i);
  double pot = pot_sum(closest, w, n);
  for (int cc = 1; cc < K; cc++) {
    /* :243-247 stable_cumsum(w * closest): chunked scan -- chunk totals first (sequential inside a
     * chunk of SCAN_CHUNK), bases by sequential sum over chunks, then base + running local sum. */
    int64_t nch = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    double base = 0.0;
    for (int64_t q = 0; q < nch; q++) {
      double loc = 0.0;
      int64_t e = (q + 1) * SCAN_CHUNK < n ? (q + 1) * SCAN_CHUNK : n;
      for (int64_t i = q * SCAN_CHUNK; i < e; i++) {
        loc = loc + w[i] * closest[i];
        cum[i] = base + loc;
      }
      base = base + loc;
    }
    int64_t cid[64];
    for (int l = 0; l < L; l++) {
      double v = u[(size_t)(cc - 1) * L + l] * pot; /* :243 */
      int64_t lo = 0, hi = n;                       /* searchsorted side='left' */
      while (lo < hi) {
        int64_t mid = (lo + hi) / 2;
        if (cum[mid] < v) lo = mid + 1;
        else hi = mid;
      }
      if (lo > n - 1) lo = n - 1; /* :249 clip */
      cid[l] = lo;
    }
    int best = 0;
    double bestpot = 0.0;
    for (int l = 0; l < L; l++) {
      double *dl = cand + (size_t)l * n;
      for (int64_t i = 0; i < n; i++) {
        double d = DIST(cid[l], i); /* :252-254 */
        dl[i] = d < closest[i] ? d : closest[i]; /* :257 np.minimum */
      }
      double p = pot_sum(dl, w, n); /* :258 */
      if (l == 0 || p < bestpot) {  /* :261 argmin */
        bestpot = p;
        best = l;
      }
    }
    pot = bestpot;
    memcpy(closest, cand + (size_t)best * n, sizeof(double) * (size_t)n);
    memcpy(C + (size_t)cc * D, Xb + cid[best] * D, sizeof(double) * D);
    if (chosen) chosen[cc] = cid[best];
  }
#undef DIST
  int rc = hamsm_set_centroids(c, s, K, C, NULL, NULL); /* counts = 0, _kmeans.py:2271 */
  free(Xb);
  free(w);
  free(xsq);
  free(closest);
  free(cand);
  free(cum);
  free(C);
  return rc;
}

/* --- a7: _kmeans.py:1620-1642 + _k_means_minibatch.pyx:59-108 */
int hamsm_minibatch_step(hamsm_ctx *c, int s, const void *X, int dt, int64_t ldx, const int32_t *rows, int64_t n_b,
                         const double *wf, int32_t *labels_out) {
  CTX(c);
  if (!c->st || s < 0 || s >= c->S) return fail(c, HAMSM_EINVAL, "bad stratum");
  stratum_t *t = &c->st[s];
  if (t->K == 0) return fail(c, HAMSM_EUNFIT, "stratum %d has no centroids", s);
  int D = c->D;
  int32_t *lab = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_b > 0 ? n_b : 1));
  double *x = (double *)malloc(sizeof(double) * (size_t)D);
  for (int64_t i = 0; i < n_b; i++) {
    load_row(x, X, dt, ldx, frame_of(rows, NULL, i), D);
    lab[i] = argmin_row(x, t, D);
  }
  for (int j = 0; j < t->K; j++) {
    double wsum = 0.0;
    for (int64_t i = 0; i < n_b; i++)
      if (lab[i] == j) wsum = wsum + (wf ? wf[frame_of(rows, NULL, i)] : 1.0); /* pyx:80-85 */
    if (wsum > 0) {
      double *cj = t->C + (size_t)j * D;
      for (int k = 0; k < D; k++) cj[k] = cj[k] * t->counts[j]; /* pyx:89-91 */
      for (int64_t i = 0; i < n_b; i++)
        if (lab[i] == j) {
          int64_t f = frame_of(rows, NULL, i);
          double wi = wf ? wf[f] : 1.0;
          for (int k = 0; k < D; k++) cj[k] = cj[k] + xval(X, dt, f * ldx + k) * wi; /* pyx:94-97 */
        }
      t->counts[j] = t->counts[j] + wsum; /* pyx:100 */
      double alpha = 1.0 / t->counts[j];  /* pyx:103 */
      for (int k = 0; k < D; k++) cj[k] = cj[k] * alpha;
      t->csq[j] = sqnorm_chain(cj, D);
    }
  }
  if (labels_out) memcpy(labels_out, lab, sizeof(int32_t) * (size_t)n_b);
  free(lab);
  free(x);
  return HAMSM_OK;
}

int hamsm_get_counts(hamsm_ctx *c, int s, double *counts) {
  CTX(c);
  if (!c->st || s < 0 || s >= c->S) return fail(c, HAMSM_EINVAL, "bad stratum");
  memcpy(counts, c->st[s].counts, sizeof(double) * (size_t)c->st[s].K);
  return HAMSM_OK;
}

/* --- _kmeans.py:1652-1673 */
int hamsm_reassign(hamsm_ctx *c, int s, const void *X, int dt, int64_t ldx, const int32_t *rows, const int32_t *cl,
                   const int64_t *brow, int n_re, double new_count) {
  CTX(c);
  if (!c->st || s < 0 || s >= c->S) return fail(c, HAMSM_EINVAL, "bad stratum");
  stratum_t *t = &c->st[s];
  for (int i = 0; i < n_re; i++) {
    if (cl[i] < 0 || cl[i] >= t->K) return fail(c, HAMSM_EINVAL, "reassign cluster out of range");
    double *cj = t->C + (size_t)cl[i] * c->D;
    if (brow) {
      load_row(cj, X, dt, ldx, frame_of(rows, NULL, brow[i]), c->D); /* :1667 centers_new[to_reassign] = X[new_centers] */
      t->csq[cl[i]] = sqnorm_chain(cj, c->D);
    }
    t->counts[cl[i]] = new_count; /* :1672 */
  }
  return HAMSM_OK;
}

/* --- batched forms: loops over the single-call restatements above ------------------------------- */
int hamsm_train_build_rows_batched(hamsm_ctx *c, const int32_t *bin, int64_t n_groups, const int64_t *goff,
                                   const int32_t *call_remap, int32_t *rows, int64_t *counts, int64_t *starts) {
  CTX(c);
  if ((counts == NULL) != (starts == NULL)) return fail(c, HAMSM_EINVAL, "train_build_rows_batched: counts and starts come together");
  int64_t *tmp = NULL;
  if (!counts) { /* (ABI 9) the caller knows them: nothing is handed back */
    tmp = (int64_t *)calloc(2 * (size_t)(c->S > 0 ? c->S : 1), sizeof(int64_t));
    if (!tmp) return fail(c, HAMSM_ENOMEM, "train_build_rows_batched");
  }
  for (int64_t g = 0; g < n_groups; g++) {
    int64_t b = goff[2 * g], e = goff[2 * g + 1];
    int64_t *cg = counts ? counts + g * c->S : tmp, *sg = starts ? starts + g * c->S : tmp + c->S;
    int rc = hamsm_train_build_rows(c, bin + b, e - b, call_remap + g * c->S, rows + b, cg, sg);
    if (rc) {
      free(tmp);
      return rc;
    }
    for (int s = 0; s < c->S; s++) sg[s] += b;
  }
  free(tmp);
  return HAMSM_OK;
}

int hamsm_kmeans_seed_multi(hamsm_ctx *c, int n_items, const int32_t *stratum, const void *X, int dt, int64_t ldx,
                            const int64_t *x_off, const int32_t *rows, const int64_t *rows_off, const int64_t *n_b,
                            const double *wf, const int64_t *sub, const int64_t *n_sub, const int32_t *K,
                            const int64_t *first, const double *u, const int32_t *L) {
  CTX(c);
  size_t esz = dt == HAMSM_F32 ? 4 : 8;
  free(c->seed_chosen);
  free(c->seed_chosen_off);
  c->seed_chosen = NULL;
  c->seed_chosen_off = (int64_t *)calloc((size_t)n_items + 1, sizeof(int64_t));
  c->seed_chosen_n = n_items > 0 ? n_items : 0;
  for (int i = 0; i < n_items; i++) c->seed_chosen_off[i + 1] = c->seed_chosen_off[i] + (K[i] > 0 ? K[i] : 0);
  c->seed_chosen = (int64_t *)calloc((size_t)(n_items > 0 ? c->seed_chosen_off[n_items] : 0) + 1, sizeof(int64_t));
  for (int i = 0; i < n_items; i++) {
    const char *Xi = (const char *)X + (size_t)x_off[i] * ldx * esz;
    const double *wi = wf ? wf + x_off[i] : NULL;
    int rc = hamsm_kmeans_seed(c, stratum[i], Xi, dt, ldx, rows + rows_off[i], n_b[i], wi, n_sub[i] ? sub : NULL, n_sub[i],
                               K[i], first[i], u, L[i], c->seed_chosen + c->seed_chosen_off[i]);
    if (rc) return rc;
    u += (size_t)(K[i] - 1) * L[i];
    sub += n_sub[i];
  }
  return HAMSM_OK;
}

int hamsm_kmeans_seed_chosen(hamsm_ctx *c, int item, int64_t *chosen, int64_t capacity) {
  CTX(c);
  if (item < 0 || item >= c->seed_chosen_n || !chosen || !c->seed_chosen) return fail(c, HAMSM_EINVAL, "seed_chosen: no such item of the last seeding call");
  if (capacity < c->seed_chosen_off[item + 1] - c->seed_chosen_off[item])
    return fail(c, HAMSM_EINVAL, "seed_chosen: item %d has %lld centroids, the buffer holds %lld", item, (long long)(c->seed_chosen_off[item + 1] - c->seed_chosen_off[item]), (long long)capacity);
  memcpy(chosen, c->seed_chosen + c->seed_chosen_off[item], sizeof(int64_t) * (size_t)(c->seed_chosen_off[item + 1] - c->seed_chosen_off[item]));
  return HAMSM_OK;
}

int hamsm_minibatch_run(hamsm_ctx *c, const void *X, int dt, int64_t ldx, const int32_t *rows, const double *wf,
                        const hamsm_step *steps, int64_t n_steps, double ratio, const int32_t *perm, int64_t n_perm_total,
                        int32_t *report, int64_t *resume, int32_t *reason) {
  CTX(c);
  size_t esz = dt == HAMSM_F32 ? 4 : 8;
  char *stopped = (char *)calloc((size_t)c->S, 1);
  for (int s = 0; s < c->S; s++) {
    resume[s] = n_steps;
    reason[s] = HAMSM_STOP_NONE;
  }
  memset(report, 0, sizeof(int32_t) * 2 * (size_t)n_steps);
  int rc = HAMSM_OK;
  for (int64_t i = 0; i < n_steps && rc == HAMSM_OK; i++) {
    const hamsm_step *st = &steps[i];
    const int s = st->stratum;
    if (s < 0 || s >= c->S) {
      rc = fail(c, HAMSM_EINVAL, "bad stratum in step %lld", (long long)i);
      break;
    }
    if (st->perm_off >= 0 && (st->n_perm < 0 || st->perm_off + st->n_perm > n_perm_total)) {
      rc = fail(c, HAMSM_EINVAL, "step %lld: permutation outside h_perm", (long long)i);
      break;
    }
    if (stopped[s]) { /* the chain of this stratum ended at an earlier step */
      if (resume[s] == n_steps) resume[s] = i;
      continue;
    }
    stratum_t *t = &c->st[s];
    const char *Xs = (const char *)X + (size_t)st->x_off * ldx * esz;
    const int32_t *rs = rows ? rows + st->rows_off : NULL;
    int zero = 0;
    for (int j = 0; j < t->K; j++) zero |= (t->counts[j] == 0.0); /* _kmeans.py:2037 */
    if (st->flags & HAMSM_STEP_FIRST) zero = 0; /* all counts are zero by construction: not a finding */
    report[2 * i] = zero;
    if (zero) {
      stopped[s] = 1;
      resume[s] = i;
      reason[s] = HAMSM_STOP_ZERO;
      continue;
    }
    rc = hamsm_minibatch_step(c, s, Xs, dt, ldx, rs, st->n_b, wf ? wf + st->x_off : NULL, NULL);
    if (rc) break;
    if (!(st->flags & 1)) continue;
    /* _kmeans.py:1643-1673 */
    double mx = t->counts[0];
    for (int j = 1; j < t->K; j++)
      if (t->counts[j] > mx) mx = t->counts[j];
    int low = 0;
    for (int j = 0; j < t->K; j++) low += (t->counts[j] < ratio * mx);
    report[2 * i + 1] = low;
    if (st->perm_off < 0) {
      if (low) {
        stopped[s] = 1;
        reason[s] = HAMSM_STOP_LOW;
      }
      continue;
    }
    if (low == 0) {
      stopped[s] = 1;
      reason[s] = HAMSM_STOP_UNUSED;
      continue;
    }
    if (low > st->n_perm || (double)low > 0.5 * (double)st->n_b) { /* :1647 */
      stopped[s] = 1;
      reason[s] = HAMSM_STOP_CAP;
      continue;
    }
    int32_t *cl = (int32_t *)malloc(sizeof(int32_t) * (size_t)low);
    int64_t *br = (int64_t *)malloc(sizeof(int64_t) * (size_t)low);
    double new_count = 0.0;
    int k = 0, have = 0;
    for (int j = 0; j < t->K; j++) {
      if (t->counts[j] < ratio * mx) {
        cl[k] = j;
        br[k] = perm[st->perm_off + k]; /* :1655 choice(n_b, replace=False, size=n) == permutation(n_b)[:n] */
        k++;
      } else if (!have || t->counts[j] < new_count) { /* :1672 np.min(weight_sums[~to_reassign]) */
        new_count = t->counts[j];
        have = 1;
      }
    }
    rc = hamsm_reassign(c, s, Xs, dt, ldx, rs, cl, br, low, new_count);
    free(cl);
    free(br);
  }
  free(stopped);
  return rc;
}

/* --- single-process stand-ins for the exchange step (tests drive N>1 through gloo on the host) */
int hamsm_comm_unique_id(char *id) {
  memset(id, 0, 128);
  return HAMSM_OK;
}
int hamsm_comm_init(hamsm_ctx *c, int nranks, int rank, const char *id) {
  CTX(c);
  (void)id;
  (void)rank;
  return nranks == 1 ? HAMSM_OK : fail(c, HAMSM_ENOTSUP, "cpu-oracle has no collective backend");
}
int hamsm_comm_destroy(hamsm_ctx *c) {
  CTX(c);
  (void)c;
  return HAMSM_OK;
}
int hamsm_allreduce_sum_f64(hamsm_ctx *c, double *b, int64_t n) {
  CTX(c);
  (void)c, (void)b, (void)n;
  return HAMSM_OK;
}
int hamsm_allreduce_min_f64(hamsm_ctx *c, double *b, int64_t n) {
  CTX(c);
  (void)c, (void)b, (void)n;
  return HAMSM_OK;
}
int hamsm_allreduce_max_f64(hamsm_ctx *c, double *b, int64_t n) {
  CTX(c);
  (void)c, (void)b, (void)n;
  return HAMSM_OK;
}
int hamsm_allreduce_sum_i64(hamsm_ctx *c, int64_t *b, int64_t n) {
  CTX(c);
  (void)c, (void)b, (void)n;
  return HAMSM_OK;
}
int hamsm_bcast(hamsm_ctx *c, void *b, size_t n, int root) {
  CTX(c);
  (void)c, (void)b, (void)n, (void)root;
  return HAMSM_OK;
}
int hamsm_gather_to_root(hamsm_ctx *c, const void *s, size_t n, void *r, const int64_t *sizes, int root) {
  CTX(c);
  (void)sizes;
  (void)root;
  if (n) memcpy(r, s, n); /* one rank */
  return HAMSM_OK;
}
int hamsm_bcast_model(hamsm_ctx *c, int root) {
  CTX(c);
  (void)c, (void)root;
  return HAMSM_OK;
}
/* (ABI 9) rows by index; the one-rank form of the grouped exchange (the oracle has no communicator: tests serve N > 1 with gloo) */
int hamsm_gather_rows(hamsm_ctx *c, const void *src, size_t row_bytes, const int64_t *idx, int64_t n, void *dst) {
  CTX(c);
  if (n < 0 || row_bytes == 0 || (row_bytes & 3)) return fail(c, HAMSM_EINVAL, "gather_rows: n = %lld rows of %zu bytes (a positive multiple of 4)", (long long)n, row_bytes);
  if (n == 0) return HAMSM_OK;
  if (!src || !idx || !dst) return fail(c, HAMSM_EINVAL, "gather_rows: null buffer");
  for (int64_t i = 0; i < n; i++) memcpy((char *)dst + (size_t)i * row_bytes, (const char *)src + (size_t)idx[i] * row_bytes, row_bytes);
  return HAMSM_OK;
}
int hamsm_exchange(hamsm_ctx *c, const void *d_send, const int64_t *h_send_bytes, void *d_recv, const int64_t *h_recv_bytes) {
  CTX(c);
  if (!h_send_bytes || !h_recv_bytes || h_send_bytes[0] != h_recv_bytes[0] || h_send_bytes[0] < 0) return fail(c, HAMSM_EINVAL, "exchange: size tables");
  if (h_send_bytes[0]) memcpy(d_recv, d_send, (size_t)h_send_bytes[0]);
  return HAMSM_OK;
}
int hamsm_bcast_strata(hamsm_ctx *c, const int32_t *roots) {
  CTX(c);
  (void)c, (void)roots; /* one rank */
  return HAMSM_OK;
}
int hamsm_timers(hamsm_ctx *c, double *out, int reset) {
  CTX(c);
  (void)c, (void)reset;
  memset(out, 0, sizeof(double) * 8);
  return HAMSM_OK;
}
int hamsm_step_kernel_ms(hamsm_ctx *c, int which, double *ms, int64_t *samples, int reset) {
  CTX(c);
  (void)which;
  (void)reset;
  if (ms) *ms = 0.0; /* no kernels here */
  if (samples) *samples = 0;
  return HAMSM_OK;
}
int hamsm_last_kernel_ms(hamsm_ctx *c, int stage, double *ms) {
  CTX(c);
  (void)c, (void)stage;
  *ms = 0.0;
  return HAMSM_OK;
}

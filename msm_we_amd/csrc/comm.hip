// comm.hip -- the exchange step of the sharded path: RCCL over xGMI, one rank per GPU.
// Frames shard by WE iteration; assignment needs no communication; the partial flux matrix, the
// cluster statistics and (for broadcast) the centroid tables are the only payloads.  Replaces the
// reference's driver-side dense sum (_fluxmatrix.py:325-327) and ray.put broadcasts
// (_clustering.py:1169-1176).
#include <rccl/rccl.h>

#include "common.h"

#define NCCLCHK(c, call)                                                                                  \
  do {                                                                                                    \
    ncclResult_t r_ = (call);                                                                             \
    if (r_ != ncclSuccess)                                                                                \
      return hamsm_fail((c), HAMSM_EDEVICE, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

static_assert(sizeof(ncclUniqueId) == 128, "hamsm_comm_unique_id hands out 128 bytes");

extern "C" int hamsm_comm_unique_id(char *id128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return HAMSM_EDEVICE;
  memcpy(id128, &id, 128);
  return HAMSM_OK;
}

extern "C" int hamsm_comm_init(hamsm_ctx *c, int nranks, int rank, const char *id128) {
  CTX(c);
  if (nranks < 1 || rank < 0 || rank >= nranks) return hamsm_fail(c, HAMSM_EINVAL, "bad rank %d/%d", rank, nranks);
  HIPCHK(c, hipSetDevice(c->device));
  if (c->comm) return hamsm_fail(c, HAMSM_EINVAL, "comm_init: a communicator exists (hamsm_comm_destroy first)");
  if (nranks == 1 && !id128) {
    c->nranks = 1;
    c->rank = 0;
    return HAMSM_OK;
  }
  if (!id128) return hamsm_fail(c, HAMSM_EINVAL, "comm_init: unique id missing");
  // a 1-rank communicator is legal and keeps every collective below on the RCCL path (used by the GPU tests)
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclComm_t comm;
  // nranks / rank are published only with a live communicator: a failed init leaves the ctx a 1-rank context
  // whose collectives are no-ops, never "nranks > 1 with a null comm"
  NCCLCHK(c, ncclCommInitRank(&comm, nranks, id, rank));
  // the device words every collective pre-check below needs are allocated HERE, with the communicator: a rank must never fail an
  // allocation between "the others have entered a collective" and "this rank enters it" 
  // (gather_to_root: 2 R + 3 words; bcast_strata: 8 + 3 S int32 -- sizes, roots, -roots; exchange: 2 R^2 + 1 words)
  const size_t want = sizeof(int64_t) * (2 * (size_t)nranks * nranks + 2 * (size_t)nranks + 8) + sizeof(int32_t) * 3 * HAMSM_MAX_STRATA;
  if (hipMalloc(&c->comm_buf, want) != hipSuccess) {
    ncclCommDestroy(comm);
    c->comm_buf = nullptr;
    return hamsm_fail(c, HAMSM_EDEVICE, "comm_init: %zu bytes of exchange scratch", want);
  }
  c->comm_buf_bytes = want;
  c->comm = (void *)comm;
  c->nranks = nranks;
  c->rank = rank;
  return HAMSM_OK;
}

extern "C" int hamsm_comm_destroy(hamsm_ctx *c) {
  CTX(c);
  if (c->comm) {
    ncclCommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
  }
  if (c->comm_buf) {
    hipFree(c->comm_buf);
    c->comm_buf = nullptr;
    c->comm_buf_bytes = 0;
  }
  c->nranks = 1;
  c->rank = 0;
  return HAMSM_OK;
}

static int allreduce(hamsm_ctx *c, void *buf, int64_t count, ncclDataType_t dt, ncclRedOp_t op) {
  if (count <= 0 || (c->nranks == 1 && !c->comm)) return HAMSM_OK;
  if (!c->comm) return hamsm_fail(c, HAMSM_EINVAL, "comm not initialised");
  NCCLCHK(c, ncclAllReduce(buf, buf, (size_t)count, dt, op, (ncclComm_t)c->comm, c->stream));
  return HAMSM_OK;
}

extern "C" int hamsm_allreduce_sum_f64(hamsm_ctx *c, double *b, int64_t n) {
  CTX(c);
  return allreduce(c, b, n, ncclDouble, ncclSum);
}
extern "C" int hamsm_allreduce_min_f64(hamsm_ctx *c, double *b, int64_t n) {
  CTX(c);
  return allreduce(c, b, n, ncclDouble, ncclMin);
}
extern "C" int hamsm_allreduce_max_f64(hamsm_ctx *c, double *b, int64_t n) {
  CTX(c);
  return allreduce(c, b, n, ncclDouble, ncclMax);
}
extern "C" int hamsm_allreduce_sum_i64(hamsm_ctx *c, int64_t *b, int64_t n) {
  CTX(c);
  return allreduce(c, b, n, ncclInt64, ncclSum);
}

extern "C" int hamsm_bcast(hamsm_ctx *c, void *b, size_t bytes, int root) {
  CTX(c);
  if (bytes == 0 || (c->nranks == 1 && !c->comm)) return HAMSM_OK;
  if (!c->comm) return hamsm_fail(c, HAMSM_EINVAL, "comm not initialised");
  NCCLCHK(c, ncclBroadcast(b, b, bytes, ncclChar, root, (ncclComm_t)c->comm, c->stream));
  return HAMSM_OK;
}

// Every rank must enter the send / recv group below with the same picture, or the ranks that did enter it wait for ever for
// the one that returned early.  So the arguments are cross-checked COLLECTIVELY first: one small max-all-reduce over
// [table, -table, complaint] -- the tables agree iff max(t_r) == -max(-t_r) for every r -- and either every rank goes on or
// every rank returns the same error.
extern "C" int hamsm_gather_to_root(hamsm_ctx *c, const void *d_send, size_t send_bytes, void *d_recv, const int64_t *h_recv_bytes,
                                    int root) {
  CTX(c);
  if (c->nranks == 1 && !c->comm) {
    if (send_bytes) HIPCHK(c, hipMemcpyAsync(d_recv, d_send, send_bytes, hipMemcpyDeviceToDevice, c->stream));
    return HAMSM_OK;
  }
  if (!c->comm) return hamsm_fail(c, HAMSM_EINVAL, "comm not initialised");
  const int R = c->nranks;
  ncclComm_t cm = (ncclComm_t)c->comm;
  int64_t complaint = 0;  // what THIS rank finds wrong with its own arguments (bit set)
  if (root < 0 || root >= R) complaint |= 1;
  if (!h_recv_bytes) complaint |= 2;
  else if ((int64_t)send_bytes != h_recv_bytes[c->rank]) complaint |= 4;
  if (h_recv_bytes)
    for (int r = 0; r < R; r++)
      if (h_recv_bytes[r] < 0) complaint |= 8;
  if (send_bytes && !d_send) complaint |= 16;
  size_t total = 0;
  if (h_recv_bytes)
    for (int r = 0; r < R; r++) total += (size_t)(h_recv_bytes[r] > 0 ? h_recv_bytes[r] : 0);
  if (c->rank == root && total && !d_recv) complaint |= 32;
  std::vector<int64_t> v(2 * R + 2);
  for (int r = 0; r < R; r++) {
    v[r] = h_recv_bytes ? h_recv_bytes[r] : 0;
    v[R + r] = -v[r];
  }
  v[2 * R] = complaint;
  v[2 * R + 1] = -(int64_t)root;  // with its twin below: every rank names the same root
  v.push_back((int64_t)root);
  // From here to the all-reduce NO rank-local return: a rank that left now would strand the others inside it.  The words live in
  // the buffer comm_init allocated; a failed upload (the device is in an error state: nothing this library can repair) is
  // remembered, the rank still enters the collective with whatever the buffer holds, and reports its own failure afterwards.
  int64_t *d_v = (int64_t *)c->comm_buf;
  const hipError_t e_up = hipMemcpyAsync(d_v, v.data(), sizeof(int64_t) * v.size(), hipMemcpyHostToDevice, c->stream);
  const ncclResult_t e_ar = ncclAllReduce(d_v, d_v, v.size(), ncclInt64, ncclMax, cm, c->stream);
  if (e_up != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "gather_to_root: upload of the argument check failed: %s", hipGetErrorString(e_up));
  if (e_ar != ncclSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "gather_to_root: ncclAllReduce failed: %s", ncclGetErrorString(e_ar));
  HIPCHK(c, hipMemcpyAsync(v.data(), d_v, sizeof(int64_t) * v.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (v[2 * R])
    return hamsm_fail(c, HAMSM_EINVAL, "gather_to_root: a rank's arguments are inconsistent (largest flag word over all ranks %lld, here %lld: "
                      "1 root, 2 no size table, 4 send size != table[rank], 8 negative size, 16 null send buffer, 32 null receive buffer "
                      "on the root)", (long long)v[2 * R],
                      (long long)complaint);
  if (v[2 * R + 2] != -v[2 * R + 1]) return hamsm_fail(c, HAMSM_EINVAL, "gather_to_root: the ranks name different roots");
  for (int r = 0; r < R; r++)
    if (v[r] != -v[R + r]) return hamsm_fail(c, HAMSM_EINVAL, "gather_to_root: the ranks hold different size tables (entry %d)", r);
  ncclResult_t first = ncclSuccess;
  const ncclResult_t gs = ncclGroupStart();
  if (send_bytes) {
    ncclResult_t r_ = ncclSend(d_send, send_bytes, ncclChar, root, cm, c->stream);
    if (first == ncclSuccess) first = r_;
  }
  if (c->rank == root) {
    size_t off = 0;
    for (int r = 0; r < R; r++) {
      if (h_recv_bytes[r]) {
        ncclResult_t r_ = ncclRecv((char *)d_recv + off, (size_t)h_recv_bytes[r], ncclChar, r, cm, c->stream);
        if (first == ncclSuccess) first = r_;
      }
      off += (size_t)h_recv_bytes[r];
    }
  }
  ncclResult_t end = gs == ncclSuccess ? ncclGroupEnd() : gs;  // (always closed, also after a failed call inside it)
  if (first != ncclSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "gather_to_root: ncclSend / ncclRecv failed: %s", ncclGetErrorString(first));
  if (end != ncclSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "gather_to_root: ncclGroupEnd failed: %s", ncclGetErrorString(end));
  return HAMSM_OK;
}

// ---- (ABI 9) the frames a rank trains on, and only those (`train_shards = "strata"`) --------------------------------------------
// d_dst row i = d_src row d_idx[i]: the sender packs, per destination rank, the rows of its shard that the destination's strata
// need (HBM-bound gather; a thread per 4-byte word, consecutive threads walk a row)
__global__ void k_gather_rows(const unsigned *__restrict__ src, int64_t wpr, const int64_t *__restrict__ idx, int64_t n_words,
                              unsigned *__restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_words; t += stride) {
    const int64_t i = t / wpr, k = t - i * wpr;
    dst[t] = src[idx[i] * wpr + k];
  }
}
extern "C" int hamsm_gather_rows(hamsm_ctx *c, const void *d_src, size_t row_bytes, const int64_t *d_idx, int64_t n, void *d_dst) {
  CTX(c);
  if (n < 0 || row_bytes == 0 || (row_bytes & 3)) return hamsm_fail(c, HAMSM_EINVAL, "gather_rows: n = %lld rows of %zu bytes (a positive multiple of 4)", (long long)n, row_bytes);
  if (n == 0) return HAMSM_OK;
  if (!d_src || !d_idx || !d_dst) return hamsm_fail(c, HAMSM_EINVAL, "gather_rows: null buffer");
  const int64_t wpr = (int64_t)(row_bytes / 4), words = n * wpr;
  const int64_t blocks = (words + 255) / 256;
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, c->stream, (const unsigned *)d_src, wpr, d_idx,
                     words, (unsigned *)d_dst);
  HIPCHK(c, hipGetLastError());
  return HAMSM_OK;
}

// Every rank sends h_send_bytes[q] bytes to rank q (consecutive in d_send, in rank order) and receives h_recv_bytes[q] bytes from
// rank q (consecutive in d_recv, in rank order): one grouped send / recv over xGMI (ncclSend / ncclRecv pairs; point-to-point
// links, so a rank's R - 1 transfers run side by side).  As in hamsm_gather_to_root the tables are cross-checked COLLECTIVELY
// first -- rank r's send row must be what the others expect to receive from it -- so that either every rank enters the group or
// every rank returns the same error; no rank-local return between the check and the group's end.
extern "C" int hamsm_exchange(hamsm_ctx *c, const void *d_send, const int64_t *h_send_bytes, void *d_recv, const int64_t *h_recv_bytes) {
  CTX(c);
  if (c->nranks == 1 && !c->comm) {
    if (!h_send_bytes || !h_recv_bytes || h_send_bytes[0] != h_recv_bytes[0] || h_send_bytes[0] < 0) return hamsm_fail(c, HAMSM_EINVAL, "exchange: size tables");
    if (h_send_bytes[0]) HIPCHK(c, hipMemcpyAsync(d_recv, d_send, (size_t)h_send_bytes[0], hipMemcpyDeviceToDevice, c->stream));
    return HAMSM_OK;
  }
  if (!c->comm) return hamsm_fail(c, HAMSM_EINVAL, "comm not initialised");
  const int R = c->nranks, me = c->rank;
  ncclComm_t cm = (ncclComm_t)c->comm;
  int64_t complaint = 0;
  if (!h_send_bytes || !h_recv_bytes) complaint |= 1;
  size_t tot_s = 0, tot_r = 0;
  for (int q = 0; q < R && !complaint; q++) {
    if (h_send_bytes[q] < 0 || h_recv_bytes[q] < 0) complaint |= 2;
    tot_s += (size_t)(h_send_bytes[q] > 0 ? h_send_bytes[q] : 0);
    tot_r += (size_t)(h_recv_bytes[q] > 0 ? h_recv_bytes[q] : 0);
  }
  if ((tot_s && !d_send) || (tot_r && !d_recv)) complaint |= 4;
  // A[r][q] = what r says it sends to q; B[r][q] = what q says it receives from r: every rank fills its own entries, the rest
  // stay -1 under the max-all-reduce; the run is consistent iff A == B
  std::vector<int64_t> v(2 * (size_t)R * R + 1, -1);
  for (int q = 0; q < R; q++) {
    v[(size_t)me * R + q] = h_send_bytes ? h_send_bytes[q] : 0;
    v[(size_t)R * R + (size_t)q * R + me] = h_recv_bytes ? h_recv_bytes[q] : 0;
  }
  v[2 * (size_t)R * R] = complaint;
  if (sizeof(int64_t) * v.size() > c->comm_buf_bytes) return hamsm_fail(c, HAMSM_EINVAL, "exchange: %d ranks", R);  // (same on every rank)
  int64_t *d_v = (int64_t *)c->comm_buf;
  const hipError_t e_up = hipMemcpyAsync(d_v, v.data(), sizeof(int64_t) * v.size(), hipMemcpyHostToDevice, c->stream);
  const ncclResult_t e_ar = ncclAllReduce(d_v, d_v, v.size(), ncclInt64, ncclMax, cm, c->stream);
  if (e_up != hipSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "exchange: upload of the argument check failed: %s", hipGetErrorString(e_up));
  if (e_ar != ncclSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "exchange: ncclAllReduce failed: %s", ncclGetErrorString(e_ar));
  HIPCHK(c, hipMemcpyAsync(v.data(), d_v, sizeof(int64_t) * v.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (v[2 * (size_t)R * R]) return hamsm_fail(c, HAMSM_EINVAL, "exchange: a rank's arguments are inconsistent (flags %lld, here %lld: 1 no table, 2 negative size, 4 null buffer)",
                                              (long long)v[2 * (size_t)R * R], (long long)complaint);
  for (size_t e = 0; e < (size_t)R * R; e++)
    if (v[e] != v[(size_t)R * R + e]) return hamsm_fail(c, HAMSM_EINVAL, "exchange: rank %zu sends %lld bytes to rank %zu, which expects %lld", e / R, (long long)v[e], e % R,
                                                        (long long)v[(size_t)R * R + e]);
  ncclResult_t first = ncclSuccess;
  const ncclResult_t gs = ncclGroupStart();
  size_t so = 0, ro = 0;
  for (int q = 0; q < R; q++) {
    if (q == me) {  // (own rows: a device copy, ordered on the same stream)
      if (h_send_bytes[q] && hipMemcpyAsync((char *)d_recv + ro, (const char *)d_send + so, (size_t)h_send_bytes[q], hipMemcpyDeviceToDevice, c->stream) != hipSuccess &&
          first == ncclSuccess)
        first = ncclUnhandledCudaError;
    } else {
      if (h_send_bytes[q]) {
        const ncclResult_t r_ = ncclSend((const char *)d_send + so, (size_t)h_send_bytes[q], ncclChar, q, cm, c->stream);
        if (first == ncclSuccess) first = r_;
      }
      if (h_recv_bytes[q]) {
        const ncclResult_t r_ = ncclRecv((char *)d_recv + ro, (size_t)h_recv_bytes[q], ncclChar, q, cm, c->stream);
        if (first == ncclSuccess) first = r_;
      }
    }
    so += (size_t)h_send_bytes[q];
    ro += (size_t)h_recv_bytes[q];
  }
  const ncclResult_t end = gs == ncclSuccess ? ncclGroupEnd() : gs;
  if (first != ncclSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "exchange: send / recv failed: %s", ncclGetErrorString(first));
  if (end != ncclSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "exchange: ncclGroupEnd failed: %s", ncclGetErrorString(end));
  return HAMSM_OK;
}

// centroids, counts and ||c||^2 of every stratum, each from the rank that trained it, to all: stratum s from rank h_root[s].
// (Strata are independent chains of partial_fits -- `_clustering.py:890-911` fits one MiniBatchKMeans per bin -- so they can be
// trained on different GPUs and the result is the single-process model bit for bit: `train_shards = "strata"`, DESIGN.md 7.  With
// every root 0 this is the "rank 0 trains" exchange of SURVEY §7 hard part 3.)  The remap table travels from rank 0.
// Collective discipline: between two collectives a rank never returns on a failure of its own -- it carries the
// failure into the next collective (a min-all-reduce of "ready" flags), after which EVERY rank returns or EVERY rank goes on;
// everything that can only fail locally and needs no partner (re-packing the fragments) comes after the last collective.
extern "C" int hamsm_bcast_strata(hamsm_ctx *c, const int32_t *h_root) {
  CTX(c);
  if (c->nranks == 1 && !c->comm) return HAMSM_OK;
  if (!c->comm) return hamsm_fail(c, HAMSM_EINVAL, "comm not initialised");
  const int S = c->S, R = c->nranks;
  ncclComm_t cm = (ncclComm_t)c->comm;
  if (sizeof(int64_t) * 8 + sizeof(int32_t) * 3 * (size_t)S > c->comm_buf_bytes) return hamsm_fail(c, HAMSM_EINVAL, "bcast_strata: %d strata", S);
  int64_t *d_flag = (int64_t *)c->comm_buf;               // [1] ready flag
  int32_t *d_K = (int32_t *)((char *)c->comm_buf + 64);   // [3 S] fitted centroids per stratum as their roots know them; roots; -roots
  // (0) every rank must name valid roots -- checked locally -- and THE SAME roots: they travel with the sizes in (1) as [root,
  // -root] under the max-all-reduce (equal on every rank iff max(root) == -max(-root), as hamsm_gather_to_root checks its root);
  // ranks that disagree would otherwise enter the grouped broadcasts of (3) with different roots and hang
  int rc_loc = h_root ? HAMSM_OK : HAMSM_EINVAL;
  for (int s = 0; s < S && !rc_loc; s++)
    if (h_root[s] < 0 || h_root[s] >= R) rc_loc = HAMSM_EINVAL;
  // (1) the strata's sizes: each root contributes its own, the others 0; one max-all-reduce.  A failed upload is carried along:
  // the rank still enters the collective.
  std::vector<int32_t> Ks(3 * (size_t)S, 0);
  for (int s = 0; s < S; s++) {
    if (!rc_loc && h_root[s] == c->rank) Ks[s] = c->st[s].K;
    Ks[S + s] = h_root ? h_root[s] : -1;
    Ks[2 * S + s] = h_root ? -h_root[s] : 1;
  }
  hipError_t e_loc = hipMemcpyAsync(d_K, Ks.data(), sizeof(int32_t) * 3 * S, hipMemcpyHostToDevice, c->stream);
  ncclResult_t e_n = ncclAllReduce(d_K, d_K, 3 * (size_t)S, ncclInt32, ncclMax, cm, c->stream);
  if (e_loc == hipSuccess && e_n == ncclSuccess) e_loc = hipMemcpyAsync(Ks.data(), d_K, sizeof(int32_t) * 3 * S, hipMemcpyDeviceToHost, c->stream);
  if (e_loc == hipSuccess) e_loc = hipStreamSynchronize(c->stream);
  bool roots_differ = false;
  for (int s = 0; s < S; s++) roots_differ = roots_differ || Ks[S + s] != -Ks[2 * S + s];
  // (2) room for them -- may fail on this rank alone (memory) -- then everybody learns whether everybody is ready
  if (!rc_loc && (e_loc != hipSuccess || e_n != ncclSuccess)) rc_loc = HAMSM_EDEVICE;
  if (!rc_loc && roots_differ) rc_loc = HAMSM_EINVAL;  // (every rank sees the same reduced words: every rank stops here)
  for (int s = 0; s < S && !rc_loc; s++) {
    if (Ks[s] < 0) rc_loc = HAMSM_EINVAL;
    else if (Ks[s] > 0 && h_root[s] != c->rank) rc_loc = hamsm_stratum_reserve(c, s, Ks[s]);
  }
  int64_t ready = rc_loc ? 0 : 1;
  const hipError_t e_up = hipMemcpyAsync(d_flag, &ready, sizeof(ready), hipMemcpyHostToDevice, c->stream);
  const ncclResult_t e_ar = ncclAllReduce(d_flag, d_flag, 1, ncclInt64, ncclMin, cm, c->stream);
  int64_t all_ready = 0;
  hipError_t e_dn = hipMemcpyAsync(&all_ready, d_flag, sizeof(all_ready), hipMemcpyDeviceToHost, c->stream);
  if (e_dn == hipSuccess) e_dn = hipStreamSynchronize(c->stream);  // (also: `ready` is a local the upload reads)
  if (rc_loc) return hamsm_fail(c, rc_loc, "bcast_strata: this rank could not prepare (%s; sizes: %s / %s; or no room for the strata)",
                                roots_differ ? "the ranks name DIFFERENT roots" : "roots", hipGetErrorString(e_loc), ncclGetErrorString(e_n));
  if (e_up != hipSuccess || e_ar != ncclSuccess || e_dn != hipSuccess)
    return hamsm_fail(c, HAMSM_EDEVICE, "bcast_strata: readiness exchange failed (%s / %s / %s)", hipGetErrorString(e_up), ncclGetErrorString(e_ar),
                      hipGetErrorString(e_dn));
  if (!all_ready) return hamsm_fail(c, HAMSM_EDEVICE, "bcast_strata: another rank could not prepare to receive the model; nothing was sent");
  for (int s = 0; s < S; s++) c->st[s].K = Ks[s];
  // (3) the tables and the remap, one group: the last collective of this call.  A failed ncclGroupStart is carried like every
  // other rank-local failure: the broadcasts are still issued (ungrouped), the error is reported after them
  ncclResult_t first = ncclSuccess;
  const ncclResult_t gs = ncclGroupStart();
  for (int s = 0; s < S; s++) {
    Stratum &t = c->st[s];
    if (t.K == 0) continue;
    const int root = h_root[s];
    ncclResult_t r3[3] = {ncclBroadcast(t.C, t.C, (size_t)t.K * c->D, ncclDouble, root, cm, c->stream),
                          ncclBroadcast(t.csq, t.csq, (size_t)t.K, ncclDouble, root, cm, c->stream),
                          ncclBroadcast(t.counts, t.counts, (size_t)t.K, ncclDouble, root, cm, c->stream)};
    for (ncclResult_t r_ : r3)
      if (first == ncclSuccess) first = r_;
  }
  {
    const ncclResult_t r_ = ncclBroadcast(c->d_remap, c->d_remap, sizeof(int32_t) * (size_t)S, ncclChar, 0, cm, c->stream);
    if (first == ncclSuccess) first = r_;
  }
  ncclResult_t end = gs == ncclSuccess ? ncclGroupEnd() : gs;  // (always closed, also after a failed call inside it)
  c->table_dirty = true;
  if (first != ncclSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "bcast_strata: ncclBroadcast failed: %s", ncclGetErrorString(first));
  if (end != ncclSuccess) return hamsm_fail(c, HAMSM_EDEVICE, "bcast_strata: ncclGroupEnd failed: %s", ncclGetErrorString(end));
  // (4) local only: re-pack the MFMA B fragments of the strata that arrived, keeping the broadcast ||c||^2
  for (int s = 0; s < S; s++) {
    Stratum &t = c->st[s];
    if (t.K == 0 || h_root[s] == c->rank) continue;
    int rc = hamsm_ensure_scratch(c, sizeof(double) * t.K);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->scratch, t.csq, sizeof(double) * t.K, hipMemcpyDeviceToDevice, c->stream));
    rc = hamsm_pack_stratum(c, s, (const double *)c->scratch);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return HAMSM_OK;
}

// centroids, counts, ||c||^2 and remap of every stratum, root -> all
extern "C" int hamsm_bcast_model(hamsm_ctx *c, int root) {
  CTX(c);
  if (c->nranks == 1 && !c->comm) return HAMSM_OK;
  std::vector<int32_t> roots((size_t)(c->S > 0 ? c->S : 1), root);
  if (root != 0 && c->comm) {
    // (the remap table of hamsm_bcast_strata travels from rank 0; a model broadcast from another root sends it from there)
    int rc = hamsm_bcast_strata(c, roots.data());
    if (rc) return rc;
    return hamsm_bcast(c, c->d_remap, sizeof(int32_t) * (size_t)c->S, root);
  }
  return hamsm_bcast_strata(c, roots.data());
}

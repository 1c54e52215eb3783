/* hamsm.h -- C ABI of the MI355X-native haMSM construction hot path.
 *
 * Drop-in boundary for the hot path of jdrusso/msm_we (reference @ /root/reference): featurized
 * frames -> stratified mini-batch k-means (train + assign) -> segment-pair flux matrix -> cluster
 * pcoord statistics -> CSR transition matrix for the host solver.  The reference has no FFI: its
 * boundary is duck-typed Python (levels B2/B3).  Each entry point below names the
 * reference interface it replaces; the ctypes binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - every function returns 0 on success or a negative HAMSM_E* code; hamsm_last_error(ctx) gives
 *    the message.  Nothing throws or longjmps across the boundary.
 *  - `d_` parameters are DEVICE addresses (from hamsm_malloc, or any HIP allocation of the same
 *    device, e.g. a torch tensor's data_ptr); `h_` parameters are HOST addresses owned by the caller.
 *  - one ctx per host thread / per GPU; a ctx is not thread-safe.
 *  - random draws are INPUTS (the host keeps numpy's legacy RandomState in the reference's call
 *    order, SURVEY.md §7 hard part 2): nothing in the library generates random numbers.
 *  - arithmetic contract: distances are csq[j] - 2*dot(x, c_j) with dot an f64 FMA
 *    chain over ascending feature index starting from +0.0; first minimum wins.  Two libraries
 *    export this ABI: csrc (HIP, gfx950; the product) and oracle/ (plain C; test infrastructure).
 */
#ifndef HAMSM_H
#define HAMSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAMSM_ABI_VERSION 10

#define HAMSM_OK 0
#define HAMSM_EINVAL (-1)   /* bad argument / call order                                         */
#define HAMSM_ENOMEM (-2)
#define HAMSM_EDEVICE (-3)  /* HIP / RCCL runtime error                                           */
#define HAMSM_EUNFIT (-4)   /* frame maps to a stratum without centroids (AssertionError at
                               stratified_clustering.py:187-189)                                 */
#define HAMSM_ERANGE (-5)   /* label outside [0, K+2) in the flux build (ValueError from
                               coo_matrix, _fluxmatrix.py:143-158)                               */
#define HAMSM_ENOTSUP (-6)

#define HAMSM_F32 0 /* frame features stored float32 (WESTPA auxdata/coord), widened exactly     */
#define HAMSM_F64 1

typedef struct hamsm_ctx hamsm_ctx;

/* ---- lifecycle ------------------------------------------------------------------------------ */
int hamsm_abi_version(void);
const char *hamsm_backend_name(void); /* "hip-gfx950" (product) or "cpu-oracle" (tests only)      */
int hamsm_create(int device, hamsm_ctx **out);
void hamsm_destroy(hamsm_ctx *ctx);
const char *hamsm_last_error(hamsm_ctx *ctx);

/* ---- device memory (the reference keeps everything in NumPy arrays; these replace ray.put of
 *      the model/coordinates, _clustering.py:1169-1176) ---------------------------------------- */
int hamsm_malloc(hamsm_ctx *ctx, size_t bytes, void **d_out);
int hamsm_free(hamsm_ctx *ctx, void *d_ptr);
int hamsm_memcpy_h2d(hamsm_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int hamsm_memcpy_d2h(hamsm_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int hamsm_memset(hamsm_ctx *ctx, void *d_dst, int byte, size_t bytes);
/* Ingest (the reference re-opens west.h5 and appends per segment, _data.py:807-932): page-locked host
 * staging buffers and uploads that overlap the host's reading / featurizing of the next block.  hamsm_memcpy_h2d_async
 * enqueues on the context's COPY stream and returns at once: the host buffer must be page-locked (hamsm_malloc_host) and
 * must not be rewritten before hamsm_copy_sync; the copied bytes are visible to later calls on the context after
 * hamsm_copy_sync (which also orders the compute stream behind the copies).                                              */
int hamsm_malloc_host(hamsm_ctx *ctx, size_t bytes, void **h_out);
int hamsm_free_host(hamsm_ctx *ctx, void *h_ptr);
int hamsm_memcpy_h2d_async(hamsm_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int hamsm_copy_sync(hamsm_ctx *ctx);
/* (ABI 9) Page-locked host blocks that belong to the PROCESS, not to a context: the destination of the discretization's
 * device-to-host copy.  The reference's post-condition is host arrays (`self.dtrajs`, `self.pair_dtrajs`,
 * `_clustering.py:1206-1240`); the caller's arrays are views of such a block and may outlive the context that filled them, so the
 * block is freed by its owner (hamsm_host_block_free), never by hamsm_destroy.  A device-to-host copy into pageable memory runs
 * at ~6 GB/s here (the runtime stages it through its own bounce buffer); into a page-locked block at the link's rate.
 * Returns HAMSM_OK / HAMSM_EINVAL / HAMSM_ENOMEM (no context: no hamsm_last_error text).                                      */
int hamsm_host_block_alloc(size_t bytes, void **h_out);
int hamsm_host_block_free(void *h_ptr);
/* float32 frames already resident in HBM widened to float64, exactly (d_dst[i] = (double)d_src[i], i < count; the buffers must
 * not overlap).  Staging starts in float32 (WESTPA's auxdata/coord, augmentation_driver.py:173-180; the reference up-casts to
 * float64 on load, _data.py:265,565) and switches to float64 in the middle of its one pass when a block holds a value float32
 * cannot carry: what was uploaded before is converted where it lies instead of being read again.  Runs on the compute stream. */
int hamsm_widen_f32(hamsm_ctx *ctx, const float *d_src, double *d_dst, int64_t count);
/* (ABI 7) d_flags[i] = 1 when row i of d_X (n rows of D values at pitch ldx, x_dtype) holds a NaN, else 0.  The reference drops /
 * zero-weights segments whose coordinates contain NaN (`_data.py:303-313, 549-553`); staging used to find them with a host scan
 * of every block (3 GB/s per core -- slower than the copy into the upload slab), now the block is scanned where it lands in HBM.
 * Compute stream.                                                                                                              */
int hamsm_nan_rows(hamsm_ctx *ctx, const void *d_X, int x_dtype, int64_t ldx, int D, int64_t n, uint8_t *d_flags);
/* (ABI 8) The host side of staging without the interpreter: n_parts byte ranges moved into the upload slab by `threads` native
 * threads (parts are cut into pieces of at most 1 MB and dealt out through one counter, so one long part and fifty short ones
 * load the threads alike).  hamsm_host_copy_parts: part i = bytes[i] bytes from h_src[i] to h_dst[i] (the iterations' arrays of
 * an in-memory source).  hamsm_host_read_parts: part i = bytes[i] bytes at offset file_off[i] of the open descriptor fd[i]
 * (pread: the run-level raw arrays of NpyDirectorySource; a short read is an error).  Replaces the reference's per-iteration
 * h5py reads + per-segment appends + concatenations (`_data.py:254-320, 531-618`); a Python thread pool doing the same copies
 * is bound by its dispatch (~60 us per job through the interpreter lock: 1.0 GB in 26-50 ms whatever the job size, measured on
 * the MI355X host).  The calls touch no device and no context state besides the error string; they may run on a thread of
 * their own while the context's owner enqueues uploads.  threads < 1 means 1; more than 64 are not started.                 */
int hamsm_host_copy_parts(hamsm_ctx *ctx, int64_t n_parts, const void *const *h_src, void *const *h_dst, const int64_t *bytes, int threads);
int hamsm_host_read_parts(hamsm_ctx *ctx, int64_t n_parts, const int32_t *fd, const int64_t *file_off, void *const *h_dst,
                          const int64_t *bytes, int threads);
int hamsm_sync(hamsm_ctx *ctx);

/* ---- model state (replaces the attributes of StratifiedClusters / modelWE that predict() reads:
 *      stratified_clustering.py:61-99, msm_we.py:301-347,398-444) ------------------------------ */
/* D features per frame (after featurize+project), pcoord dimensionality, number of WE bins.
 * D <= 128: the kernels keep a frame's features as 4-wide MFMA k-steps in registers (instantiated for up to 32 k-steps).
 * 128 < D <= 16384: the generic kernels (run-time k-step count; no f32 ranking pass, no persistent launch) -- same results,
 * lower rates.  Beyond: HAMSM_ENOTSUP.  k-means++ seeding of a batch of more than 3072 rows keeps its
 * candidates' rows in LDS: L * (D + 1057) * 8 bytes must fit 160 KB (L = 2 + ln K candidates), else HAMSM_ENOTSUP.         */
int hamsm_set_geometry(hamsm_ctx *ctx, int D, int pcoord_ndim, int n_strata);
/* RectilinearBinMapper: per dimension n_bounds[d] float32 boundaries, concatenated in h_bounds.
 * bin_mapper.assign call sites: stratified_clustering.py:134, _clustering.py:877.                */
int hamsm_set_mapper_rectilinear(hamsm_ctx *ctx, const int32_t *h_n_bounds, const float *h_bounds);
/* VoronoiBinMapper with the Euclidean distance function (westpa: bin = argmin_c dfunc(p, centers); reference call sites
 * as above, `find_nearest_bin` _clustering.py:1357-1396): n_centers == n_strata float32 centres [n][ndim]; the pcoord is cast
 * to float32 like westpa's coord_dtype, squared distances are accumulated in float64 over ascending dimension, the first
 * minimum wins.  westpa's source is not in the build image: restated from its documented behaviour, parity unpinned.       */
int hamsm_set_mapper_voronoi(hamsm_ctx *ctx, int n_centers, const float *h_centers);
/* Any other mapper (a Voronoi mapper with a user distance function, FuncBinMapper, recursive mappers ...): its `assign`
 * is Python and runs on the host; the library then takes the bin of every frame from an array instead of computing it.
 * After hamsm_set_mapper_external, hamsm_set_frame_bins names that array (int32, device, -1 = outside the bin space) for the
 * NEXT call of hamsm_classify / hamsm_assign / hamsm_predict_host / hamsm_train_prepass: element i is the bin of the frame
 * whose pcoord is d_pcoord[i] of that call (hamsm_predict_host: h_bins is a HOST array).                                   */
int hamsm_set_mapper_external(hamsm_ctx *ctx);
int hamsm_set_frame_bins(hamsm_ctx *ctx, const int32_t *d_bins);
int hamsm_set_frame_bins_host(hamsm_ctx *ctx, const int32_t *h_bins, int64_t n);
/* we_remap table (stratified_clustering.py:85,135; written at _clustering.py:730,1078).          */
int hamsm_set_remap(hamsm_ctx *ctx, const int32_t *h_remap);
/* basis / target pcoord boxes [ndim][2], strict inequalities (msm_we.py:462-527).                */
int hamsm_set_bounds(hamsm_ctx *ctx, const double *h_basis, const double *h_target);
/* cluster_models[stratum].cluster_centers_ (get/SET: cleaning deletes rows, _clustering.py:1041).
 * h_csq: ||c||^2 as the caller computed it (sklearn: einsum, _k_means_lloyd.pyx:96) or NULL to
 * let the library compute it (FMA chain).  h_counts: MiniBatchKMeans._counts or NULL (zeros).
 * K_s == 0 marks the stratum unfitted (no cluster_centers_ attribute).                           */
int hamsm_set_centroids(hamsm_ctx *ctx, int stratum, int K_s, const double *h_C,
                        const double *h_csq, const double *h_counts);
int hamsm_get_centroids(hamsm_ctx *ctx, int stratum, double *h_C, double *h_counts, double *h_csq);
int hamsm_num_centroids(hamsm_ctx *ctx, int stratum); /* >=0, or negative error                    */

/* ---- a3/a4: stratum id + basis/target flags per frame ---------------------------------------
 * d_bin[i]  = mapper.assign(p_i) (then we_remap if apply_remap)           (int32)
 * d_flags[i]= bit0 basis(p_i) | bit1 target(p_i)                          (uint8)                 */
int hamsm_classify(hamsm_ctx *ctx, const double *d_pcoord, int64_t n, int apply_remap,
                   int32_t *d_bin, uint8_t *d_flags);

/* ---- a2: the linear map of the dimensionality reduction (reduceCoordinates, _dimensionality.py:37-66, called per
 * iteration at _clustering.py:894-896 and 1291-1296): what sklearn's PCA / IncrementalPCA .transform computes,
 *   d_Y[i][j] = (sum_k x[i][k] W[j][k]) - (sum_k mean[k] W[j][k])   ( / h_scale[j] when whitening ),
 * every sum an in-order fma chain over k.  h_W is components_ [d][Draw]; h_mean [Draw] and h_scale [d] are nullable;
 * d_X holds n rows of Draw raw features (x_dtype, row stride ldx); d_Y is [n][d] float64.  Needs no geometry.      */
int hamsm_project(hamsm_ctx *ctx, const void *d_X, int x_dtype, int64_t ldx, int64_t n, int Draw,
                  const double *h_mean, const double *h_W, int d, const double *h_scale, double *d_Y);

/* ---- f3: streaming PCA FIT by covariance accumulation.  The reference fits scikit-learn's IncrementalPCA one
 * iteration at a time (`do_pca` / `do_full_pca`, _dimensionality.py:69-108, driven by :171-241: a forked process and an SVD of
 * (components + batch) x features per iteration).  These three calls replace that loop by one pass over the same features:
 *   hamsm_pca_begin(D)                      zero the sums
 *   hamsm_pca_accumulate(d_X, ...) ...      any number of blocks of n rows x D raw features (x_dtype, row stride ldx)
 *   hamsm_pca_finish(&n, shift, S1, S2)     n = rows seen; shift[D] = the first row seen; S1[a] = sum_f z_a(f);
 *                                           S2[a][b] = sum_f z_a(f) z_b(f)  (full symmetric [D][D]),  z(f) = x(f) - shift
 * from which the caller forms mean = shift + S1/n, cov = (S2 - S1 S1^T / n) / (n - 1) and its eigenvectors (the D x D
 * eigen-solve is host work, msm_we_amd/pca.py).  Opt-in (`modelWE.pca_fit = "device"`): IncrementalPCA with a component cap
 * truncates at every batch, so its result is not a function of the covariance -- the default fit stays the reference's own.
 * Order (bit-identical across the two libraries): per accumulate call the n rows are cut into G strips of L rows,
 *   Dp = 32 ceil((D+1)/32),  G = max(1, min(8192, ceil(n/512), floor(2^25 / Dp^2))),  L = 4 ceil(ceil(n/G)/4);
 * per strip and pair a <= b the products are accumulated in row order as one fma chain from 0; the strips' sums are added in
 * strip order inside groups of 64 consecutive strips, the groups' sums in group order, and that total is added to the
 * running sum.  Needs no geometry.                                                                                     */
int hamsm_pca_begin(hamsm_ctx *ctx, int Draw);
int hamsm_pca_accumulate(hamsm_ctx *ctx, const void *d_X, int x_dtype, int64_t ldx, int64_t n);
int hamsm_pca_finish(hamsm_ctx *ctx, int64_t *n_total, double *h_shift, double *h_S1, double *h_S2);

/* ---- a1: the distance-to-reference featurizer the reference ships with its tests (`processCoordinates`,
 * tests/test_msm_we.py:19-36: MDAnalysis `distances.dist(frame_atoms, reference_atoms)[2]` per frame; contract
 * docs/usage.rst:41-60).  For frame i and selected atom a
 *   d_X[i][a] = sqrt(dx*dx + dy*dy + dz*dz),   d? = (double)((float)coord[i][h_idx[a]][?] - h_ref[a][?])
 * i.e. positions are float32, their difference is taken in float32, squares / sum (x, y, z in this order) / root in
 * float64, nothing fused -- what MDAnalysis' calc_bonds kernel computes without a box.  d_coords holds n frames of
 * n_atoms x 3 coordinates (c_dtype; float64 input is rounded to float32 first, as loading it into a Universe does),
 * frame stride ldc elements; h_idx [n_sel] atom indices; h_ref [n_sel][3] float32 reference positions of the SELECTED
 * atoms; d_X [n][n_sel] float64, row stride ldx >= n_sel.  Needs no geometry.  (The flattening featurizer of
 * examples/hamsm_construction.ipynb needs no entry point: pass the coordinates as features with D = 3 n_atoms.)
 * MDAnalysis is not in the build image: this entry point is pinned to its published algorithm, not to its output. */
int hamsm_featurize_dist(hamsm_ctx *ctx, const void *d_coords, int c_dtype, int64_t ldc, int64_t n, int n_atoms,
                         const int32_t *h_idx, int n_sel, const float *h_ref, double *d_X, int64_t ldx);

/* ---- a9: StratifiedClusters.predict (stratified_clustering.py:101-212) fused with a3/a4 -------
 * label = Kfit+1 if target(p) else Kfit if basis(p) else off(s) + argmin_j ||x - c_j||^2 over
 * the centroids of s = remap[bin(p)].  d_seen (nullable): int32[2][n_strata], [0][s]=1 when a
 * target frame was seen in stratum s, [1][s]=1 for basis (target_bins / basis_bins sets, :159-169).
 * ldx = row stride of d_X in elements (>= D).
 * The argmin is the one of the arithmetic contract above (f64 fma chains, first index on ties).  The HIP library reaches
 * it in two passes: the centroids are ranked on the matrix unit in reduced precision (f32 operands as three exact bf16
 * pieces, or plain f32) and a label is final only when a proven bound on the difference to the f64 values separates the
 * two best candidates; every other frame (near ties, NaN / inf, f32 overflow) is
 * decided by the exact f64 kernel.  HAMSM_ASSIGN_FILTER=0 in the environment runs the exact kernel alone.        */
int hamsm_assign(hamsm_ctx *ctx, const void *d_X, int x_dtype, int64_t ldx, const double *d_pcoord,
                 int64_t n, int32_t *d_labels, int32_t *d_seen);
/* (ABI 10) Both sides of the same n segments in one call -- what one task of the reference does for an iteration
 * (`do_ray_discretization` predicts the parent frames, then the child frames: _clustering.py:1299-1301, :1315-1316):
 * hamsm_assign(d_Xp, d_p0 -> d_parent) followed by hamsm_assign(d_Xc, d_p1 -> d_child), both marking d_seen, with the same
 * results and errors, but enqueued back to back and checked once (one error read-back and one idle gap fewer: 0.05 ms per
 * pair at 10 M segments on an MI355X).  Not with an external mapper (its frame bins are set per side: HAMSM_EINVAL).       */
int hamsm_assign_pair(hamsm_ctx *ctx, const void *d_Xp, const void *d_Xc, int x_dtype, int64_t ldx, const double *d_p0,
                      const double *d_p1, int64_t n, int32_t *d_parent, int32_t *d_child, int32_t *d_seen);
/* The discretization as the reference hands it out (`dtrajs`: child labels, `pair_dtrajs`: (parent, child) per segment, both
 * int64 -- _clustering.py:1226-1240): widened on the device and copied to the caller's HOST arrays h_child[n] and
 * h_pairs[n][2] (either may be NULL).                                                                                      */
int hamsm_labels_to_host(hamsm_ctx *ctx, const int32_t *d_parent, const int32_t *d_child, int64_t n, int64_t *h_child,
                         int64_t *h_pairs);
/* Same, host buffers in and out (the drop-in `predict(coords)` call of B3).                      */
int hamsm_predict_host(hamsm_ctx *ctx, const void *h_X, int x_dtype, const double *h_pcoord,
                       int64_t n, int32_t *h_labels, int32_t *h_seen);

/* ---- a10: flux matrix (FluxMatrixMixin.build_flux_matrix, _fluxmatrix.py:97-164) --------------
 * F[(a,b)] += w with a = parent label, b = child label, then b=K+1 if target(p1); a=K if
 * basis(p0); b=K if basis(p1) (that order, :135-137).  d_F is (K+2)^2 f64, caller-zeroed.
 * mode 0: f64 atomics (order-free, within 1e-6 rel of the reference); mode 1: deterministic,
 * bit-identical to summing per-iteration dense matrices in ascending iteration order
 * (needs d_iter_offsets).  d_iter_offsets (int64[n_iters+1], device, nullable): when given, the
 * segments processed are [offs[0], offs[n_iters]) RELATIVE TO THE ARRAY BASES and iteration t is
 * [offs[t], offs[t+1]); n must equal offs[n_iters]-offs[0].  When NULL, segments [0, n) form one
 * iteration.                                                                                     */
int hamsm_flux_accumulate(hamsm_ctx *ctx, const int32_t *d_parent, const int32_t *d_child,
                          const double *d_p0, const double *d_p1, const double *d_w, int64_t n,
                          int K, double *d_F, int mode, const int64_t *d_iter_offsets,
                          int64_t n_iters);
/* fluxMatrix / nI (_fluxmatrix.py:342), elementwise true division.                               */
int hamsm_scale(hamsm_ctx *ctx, double *d_F, int64_t count, double divisor);

/* ---- a11: per-cluster pcoord statistics (ClusteringMixin.get_cluster_centers,
 *      _clustering.py:1559-1591): over frames with label c: sum, min, max of pcoord dim 0 and
 *      count.  All outputs length nstates, device, caller-initialised (0, +inf, -inf, 0).
 *      Sums are accumulated in ascending frame order per cluster (deterministic).               */
int hamsm_cluster_stats(hamsm_ctx *ctx, const int32_t *d_labels, const double *d_pcoord, int64_t n,
                        int nstates, double *d_sum, double *d_min, double *d_max, int64_t *d_count);

/* ---- f4: segments grouped by cluster (the gather of ClusteringMixin.update_cluster_structures, _clustering.py:1398-1526:
 *      "for every segment, append its structure to its cluster's list", i.e. a stable grouping by label).  d_order[k] (int32,
 *      device, length n) receives the segment indices sorted by label, segments of one label in ascending order;
 *      h_indptr[nstates + 1] (host) the offsets of each label's run in it.  Labels outside [0, nstates) -> HAMSM_ERANGE.   */
int hamsm_group_by_label(hamsm_ctx *ctx, const int32_t *d_labels, int64_t n, int nstates, int32_t *d_order,
                         int64_t *h_indptr);

/* ---- a12 on the device (organize_stratified, _clustering.py:920-1142): the flux matrix never has to visit the host.
 * hamsm_pattern_csr: the sparsity pattern of a dense n x n matrix as CSR (h_indptr[n+1], h_indices[cap]) -- the input of the
 *   strongly-connected-components search (utils.py:21-84).  h_indices == NULL: count only (*h_nnz and h_indptr are filled).
 * hamsm_np_sum: what `np.sum(a)` returns for `count` doubles laid out contiguously in a's MEMORY order, bit for bit: numpy's
 *   reduction walks memory in chunks of its buffer size (8192 elements), sums every chunk pairwise (blocks of <= 128 with 8
 *   interleaved partial sums, larger blocks split at n/2 rounded down to a multiple of 8) and adds the chunk sums in order --
 *   the divisor of `fluxMatrix / np.sum(fluxMatrix)` (:1121).  (That fluxMatrix is the result of `a[perm, :][:, perm]`, which
 *   numpy lays out column-major: the sum is taken over the TRANSPOSED permuted matrix, hamsm_permute_scale(transpose = 1).)
 * hamsm_permute_scale: d_out[i][j] = d_F[perm[i]][perm[j]] / divisor  (`fluxMatrix[perm][:, perm] / sum`, :1117-1121;
 *   divisor 1.0 = permutation only; transpose != 0: d_out[i][j] = d_F[perm[j]][perm[i]] / divisor).  d_out must not alias d_F. */
int hamsm_pattern_csr(hamsm_ctx *ctx, const double *d_F, int nstates, int64_t *h_indptr, int32_t *h_indices, int64_t cap,
                      int64_t *h_nnz);
int hamsm_np_sum(hamsm_ctx *ctx, const double *d_A, int64_t count, double *h_out);
int hamsm_permute_scale(hamsm_ctx *ctx, const double *d_F, int nstates, const int32_t *h_perm, double *d_out, double divisor,
                        int transpose);

/* ---- a13 feed: row-normalised transition matrix as CSR (AnalysisMixin.get_Tmatrix,
 *      _analysis.py:23-79): zero rows -> self loop, target rows -> uniform over basis.
 *      d_F is nstates^2 (already organised).  Outputs are HOST arrays: h_indptr[nstates+1],
 *      h_indices/h_data capacity `cap`; *h_nnz receives the count (returns HAMSM_ENOMEM if > cap).
 *      h_indices == NULL and h_data == NULL: count only -- h_indptr and *h_nnz are filled, so that the caller can size the
 *      arrays exactly instead of reserving nstates^2 entries.                                                           */
int hamsm_tmatrix_csr(hamsm_ctx *ctx, const double *d_F, int nstates, const int32_t *h_basis, int nb,
                      const int32_t *h_target, int nt, int64_t *h_indptr, int32_t *h_indices,
                      double *h_data, int64_t cap, int64_t *h_nnz);

/* ---- a5: training-call planning support (do_stratified_clustering, _clustering.py:843-886) ----
 * d_bin[i] = mapper.assign(p_i) WITHOUT we_remap, or -1 when basis(p_i) or target(p_i)
 * (:872-877).  h_hist[it][s] = number of frames of iteration `it` in bin s (rows of
 * h_iter_offsets int64[n_iters+1] delimit iterations).                                           */
int hamsm_train_prepass(hamsm_ctx *ctx, const double *d_pcoord, int64_t n,
                        const int64_t *h_iter_offsets, int64_t n_iters, int32_t *d_bin,
                        int64_t *h_hist);
/* Row lists of one training call over frames [0,n) of d_bin: the reference computes indices in the
 * basis/target-FILTERED list and applies them to the UNFILTERED coordinates (quirk Q1,
 * _clustering.py:872-877 vs :892-899); d_rows receives, bin by bin (ascending), the ascending
 * filtered ranks of the frames whose h_call_remap[bin] equals that bin.  h_counts[s] = rows of
 * bin s; h_starts[s] = offset of its list in d_rows.                                             */
int hamsm_train_build_rows(hamsm_ctx *ctx, const int32_t *d_bin, int64_t n,
                           const int32_t *h_call_remap, int32_t *d_rows, int64_t *h_counts,
                           int64_t *h_starts);

/* ---- a6: k-means++ seeding (sklearn _kmeans.py:174-276, via _init_centroids :955-1035) --------
 * Batch = rows d_rows[0..n_b) of d_X (NULL d_rows = identity); d_w per-FRAME weights indexed like
 * d_X (NULL = ones).  h_sub/n_sub: the randint(0,n,init_size) subsample (:1006-1011) or NULL.
 * first = random_state.choice(n, p=w/sum w) drawn by the host; h_u = uniform(size=L) draws for
 * seeds 1..K_s-1 in call order, L = 2+int(ln K_s).  Result becomes the stratum's centroids with
 * counts = 0.  h_chosen (nullable) receives the K_s chosen batch indices.                        */
int hamsm_kmeans_seed(hamsm_ctx *ctx, int stratum, const void *d_X, int x_dtype, int64_t ldx,
                      const int32_t *d_rows, int64_t n_b, const double *d_w, const int64_t *h_sub,
                      int64_t n_sub, int K_s, int64_t first, const double *h_u, int L,
                      int64_t *h_chosen);

/* ---- a7: one MiniBatchKMeans step on a batch (sklearn _kmeans.py:1620-1642:
 *      _labels_inertia + _minibatch_update_dense, _k_means_minibatch.pyx:59-108) -----------------
 * assign the batch to the stratum's centroids, then per cluster, in ascending sample order:
 * c*=count; c+=x*w...; count+=wsum; c*=1/count.  d_labels (nullable) receives the batch labels.  */
int hamsm_minibatch_step(hamsm_ctx *ctx, int stratum, const void *d_X, int x_dtype, int64_t ldx,
                         const int32_t *d_rows, int64_t n_b, const double *d_w, int32_t *d_labels);
/* counts of a stratum (for the host-side low-count test, _kmeans.py:1643-1650).                  */
int hamsm_get_counts(hamsm_ctx *ctx, int stratum, double *h_counts);
/* random reassignment (_kmeans.py:1652-1673): centroid h_cluster[i] := batch row h_batch_row[i];
 * counts[h_cluster[i]] := new_count.  The choice() draw is the host's.                           */
int hamsm_reassign(hamsm_ctx *ctx, int stratum, const void *d_X, int x_dtype, int64_t ldx,
                   const int32_t *d_rows, const int32_t *h_cluster, const int64_t *h_batch_row,
                   int n_re, double new_count);

/* ---- a5-a7 batched: the whole training pass without per-step host round trips ------------------
 * The reference runs one partial_fit per WE bin per training call, each behind a fork and a pickle
 * (_clustering.py:664-691).  The grouping of iterations into training calls depends only on the
 * prepass histogram, so the host plans every call up front; these entry points then execute the plan.
 *
 * hamsm_train_build_rows_batched: hamsm_train_build_rows for n_groups calls in one launch.  Group g
 * covers frames [h_group_range[2g], h_group_range[2g+1]) of d_bin (disjoint ranges) and uses
 * h_call_remap[g*S .. g*S+S).  Its row lists are written to d_rows starting at h_group_range[2g]
 * (ranks stay relative to the group's first frame); h_counts[g*S+s] rows of bin s start at
 * d_rows + h_starts[g*S+s].                                                                         (ABI 9) h_counts and h_starts may both be NULL: the caller derives them from the
 * prepass histogram (counts = the histogram pushed through the call's remap; starts = group begin + running sum over the
 * strata) and the call returns with the kernel enqueued instead of waiting for the read-back.                               */
int hamsm_train_build_rows_batched(hamsm_ctx *ctx, const int32_t *d_bin, int64_t n_groups,
                                   const int64_t *h_group_range, const int32_t *h_call_remap,
                                   int32_t *d_rows, int64_t *h_counts, int64_t *h_starts);

/* k-means++ for several strata at once (they are independent; one workgroup each, concurrently).
 * Arrays of length n_items; per item the arguments of hamsm_kmeans_seed, with batch rows
 * d_rows + rows_off[i] indexing frames of d_X + x_off[i]*ldx.  h_u holds the items' uniform draws
 * back to back ((K_s[i]-1)*L[i] each); h_sub likewise (n_sub[i] each, 0 = no subsample).           */
int hamsm_kmeans_seed_multi(hamsm_ctx *ctx, int n_items, const int32_t *h_stratum, const void *d_X,
                            int x_dtype, int64_t ldx, const int64_t *h_x_off, const int32_t *d_rows,
                            const int64_t *h_rows_off, const int64_t *h_n_b, const double *d_w,
                            const int64_t *h_sub, const int64_t *h_n_sub, const int32_t *h_K,
                            const int64_t *h_first, const double *h_u, const int32_t *h_L);

/* (ABI 7) The batch indices item `item` of the LAST hamsm_kmeans_seed_multi call chose, in order (h_chosen[K_s]; index into the
 * subsample when one was given) -- what hamsm_kmeans_seed returns through its own h_chosen.  The parity tests compare them
 * with the reference's `_kmeans_plusplus` indices (sklearn _kmeans.py:214-276, returned as `indices`); the product never
 * calls it (it synchronises).  `capacity` (ABI 9): elements h_chosen can hold -- fewer than the item's K_s is an error, as is a
 * call after another entry point has taken the scratch arena the indices live in.                                          */
int hamsm_kmeans_seed_chosen(hamsm_ctx *ctx, int item, int64_t *h_chosen, int64_t capacity);

/* One planned minibatch step: batch rows d_rows[rows_off .. rows_off+n_b) (relative to frame x_off)
 * update the centroids of `stratum`.
 * flags bit1 (HAMSM_STEP_FIRST): the first step after seeding -- every count is zero by construction (_kmeans.py:2271),
 * which is what makes `_random_reassign` fire on it (:2037); the zero-count stop (HAMSM_STOP_ZERO) is not applied.
 * flags bit0: the low-count rule (_kmeans.py:1643-1673) is evaluated after the step (the host's closed-form
 * `_random_reassign` schedule, :2029-2043).  The rule needs `random_state.choice(n_b, replace=False,
 * size=n_low)`, i.e. the first n_low entries of `random_state.permutation(n_b)` -- random numbers are consumed
 * only when n_low > 0, so the host PREDICTS per step whether they will be:
 *   perm_off <  0 : predicted n_low == 0, nothing drawn;
 *   perm_off >= 0 : predicted n_low > 0: h_perm[perm_off .. perm_off+n_perm) holds the head of the permutation
 *                   the host drew; the device re-seeds the low centroids from it (ascending centroid index
 *                   <-> permutation order, counts := min over the others), exactly as :1652-1673.            */
#define HAMSM_STEP_TEST 1
#define HAMSM_STEP_FIRST 2
typedef struct hamsm_step {
  int32_t stratum;
  int32_t flags;
  int64_t x_off;
  int64_t rows_off;
  int64_t n_b;
  int64_t perm_off;
  int32_t n_perm;
  int32_t n_span; /* optional: the step's row indices are all below this (the frames of its WE iterations); 0 = not stated.
                     Lets the library run a large call's E-step through the assignment kernels (labels per frame of the span) */
} hamsm_step;

/* why a stratum's chain of steps stopped (h_reason of hamsm_minibatch_run) */
#define HAMSM_STOP_NONE 0   /* every step applied, every prediction held                                   */
#define HAMSM_STOP_LOW 1    /* last applied step: n_low > 0 but no permutation (host must re-seed, then resume) */
#define HAMSM_STOP_UNUSED 2 /* last applied step: a permutation was supplied but n_low == 0 (host rewinds its RNG) */
#define HAMSM_STOP_CAP 3    /* last applied step: n_low > n_perm or n_low > 0.5 n_b (:1647-1651): host handles it  */
#define HAMSM_STOP_ZERO 4   /* a count was zero before step h_resume[s] (`(self._counts == 0).any()`, :2037):
                               the host's schedule does not hold from here                                     */

/* Execute steps in order (steps of different strata may overlap; steps of one stratum never do)
 * with no host involvement, each stratum until one of its predictions fails.  h_resume[s] = index of
 * the first step of stratum s that was NOT applied (n_steps when all were), h_reason[s] = HAMSM_STOP_*.
 * h_report[2*i] = 1 if some count was 0 before step i; h_report[2*i+1] = n_low after step i (flags bit0),
 * whether or not it was acted upon.                                                                   */
int hamsm_minibatch_run(hamsm_ctx *ctx, const void *d_X, int x_dtype, int64_t ldx,
                        const int32_t *d_rows, const double *d_w, const hamsm_step *h_steps,
                        int64_t n_steps, double reassignment_ratio, const int32_t *h_perm,
                        int64_t n_perm_total, int32_t *h_report, int64_t *h_resume, int32_t *h_reason);

/* ---- multi-GPU: frames shard by WE iteration; these are the only exchanges.
 *      Replaces the driver-side sum of per-task flux matrices (_fluxmatrix.py:325-327) and
 *      ray.put broadcasts (_clustering.py:1169-1176).  RCCL over xGMI in the product.           */
int hamsm_comm_unique_id(char *h_id128);
/* nranks == 1 with h_id128 == NULL: no communicator, the collectives below are no-ops.  With an id a communicator is
 * created even for one rank (the collectives then run through RCCL and are the identity). */
int hamsm_comm_init(hamsm_ctx *ctx, int nranks, int rank, const char *h_id128);
int hamsm_comm_destroy(hamsm_ctx *ctx);
int hamsm_allreduce_sum_f64(hamsm_ctx *ctx, double *d_buf, int64_t count);
int hamsm_allreduce_min_f64(hamsm_ctx *ctx, double *d_buf, int64_t count);
int hamsm_allreduce_max_f64(hamsm_ctx *ctx, double *d_buf, int64_t count);
int hamsm_allreduce_sum_i64(hamsm_ctx *ctx, int64_t *d_buf, int64_t count);
int hamsm_bcast(hamsm_ctx *ctx, void *d_buf, size_t bytes, int root);
/* every rank's block to the root, rank by rank: rank r's send_bytes bytes land at d_recv + sum of h_recv_bytes[0..r) on the
 * root (h_recv_bytes[nranks], the same on every rank; d_recv is only read on the root).  Used to bring the training frames
 * of every iteration shard to the rank that trains ("exact" mode over ALL iterations): point-to-point
 * sends over xGMI inside one RCCL group.  One rank: a device copy.                                                  */
int hamsm_gather_to_root(hamsm_ctx *ctx, const void *d_send, size_t send_bytes, void *d_recv, const int64_t *h_recv_bytes,
                         int root);
/* broadcast the whole model state (centroids, counts, csq, remap) from root to every rank.       */
int hamsm_bcast_model(hamsm_ctx *ctx, int root);
/* (ABI 9) `train_shards = "strata"` without shipping every frame to every rank (round 4 did: R rooted gathers, each rank
 * received ALL training frames).  The sender packs, per destination, the rows of its shard whose stratum -- after the training
 * call's remap, `_clustering.py:800-826` -- trains on that rank (hamsm_gather_rows: d_dst row i = d_src row d_idx[i], rows of
 * row_bytes = a multiple of 4 bytes), and ONE grouped exchange moves them (hamsm_exchange: h_send_bytes[q] bytes to rank q,
 * consecutive in d_send in rank order; h_recv_bytes[q] bytes from rank q, consecutive in d_recv in rank order; the size tables
 * are cross-checked collectively before any transfer, as in hamsm_gather_to_root).  Ranks hold consecutive blocks of WE
 * iterations, so what arrives in rank order is in iteration order: the receiving rank trains its strata on exactly the rows, in
 * exactly the order, one process holding every iteration would (`_clustering.py:843-899`).                                     */
int hamsm_gather_rows(hamsm_ctx *ctx, const void *d_src, size_t row_bytes, const int64_t *d_idx, int64_t n, void *d_dst);
int hamsm_exchange(hamsm_ctx *ctx, const void *d_send, const int64_t *h_send_bytes, void *d_recv, const int64_t *h_recv_bytes);
/* (ABI 7) the same with a root PER STRATUM: stratum s's centroids, counts and csq travel from rank h_root[s] (h_root[S], the same
 * on every rank), the remap table from rank 0.  The reference fits one MiniBatchKMeans per WE bin (`_clustering.py:890-911`,
 * `stratified_clustering.py:55-99`): the strata's chains of partial_fits are independent, so different ranks can train
 * different strata and the model is the single-process one bit for bit (`train_shards = "strata"`).                    */
int hamsm_bcast_strata(hamsm_ctx *ctx, const int32_t *h_root);

/* ---- timers: seconds of device time per stage since the last reset, measured with HIP events on
 *      the library's stream.  h_out[8]: assign, flux, stats, seed, step, reassign, prepass, other */
int hamsm_timers(hamsm_ctx *ctx, double *h_out, int reset);
/* mean duration (ms) of the two training kernels, sampled at every 128th launch pair on the context's queue (sampling starts
 * with the first call of this function on a context) since
 * the last reset: which = 0 the E-step (k_step_assign), 1 the M-step (k_step_update) -- event to event, i.e. including the
 * launch boundary --, 2 the E-step by the device's 100 MHz clock (first wave in to last wave out: what a kernel trace reports);
 * *h_samples (nullable) = pairs sampled.                                                                                   */
int hamsm_step_kernel_ms(hamsm_ctx *ctx, int which, double *h_ms, int64_t *h_samples, int reset);
/* last kernel-only duration (ms) of the named stage: 0 assign, 1 flux, 4 step                    */
int hamsm_last_kernel_ms(hamsm_ctx *ctx, int stage, double *h_ms);

#ifdef __cplusplus
}
#endif
#endif /* HAMSM_H */

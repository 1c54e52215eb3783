"""``ClusteringMixin``: stratified k-means training, discretization and cleaning.

Mirrors the stratified branch of the reference's ``msm_we/_hamsm/_clustering.py`` (``:142-196``,
``:525-1142``, ``:1144-1396``, ``:1528-1611``): same method names, arguments, post-conditions and
quirks, with the arithmetic in the HIP library.  The reference's
fork-per-iteration / Ray task machinery (``:672-691``, ``:1183-1233``) has no counterpart: frames
are HBM-resident and one launch covers every segment.
"""
from __future__ import annotations

import logging
import numbers
import os
import sys
import time
from copy import deepcopy

import numpy as np

from . import _abi
from .data import DeviceStore
from .stratified_clustering import StratifiedClusters

log = logging.getLogger("msm_we_amd")

__all__ = ["ClusteringMixin", "find_connected_sets"]


def find_connected_sets(C, directed=True):
    """Strongly connected components, largest first, each sorted (reference ``utils.py:21-84``;
    host-side by design)."""
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csgraph

    C = np.asarray(C)
    n = C.shape[0]
    B = C != 0  # sparsity pattern as CSR without scipy's dense -> COO pass (2x the time of the component search itself)
    idx = np.flatnonzero(B)
    indptr = np.concatenate([[0], np.cumsum(B.sum(axis=1))]).astype(np.int32)
    Cs = sp.csr_matrix((np.ones(len(idx), dtype=np.int8), (idx % n).astype(np.int32), indptr), shape=C.shape)
    nc, lab = csgraph.connected_components(Cs, directed=directed, connection="strong")
    order = np.argsort(lab)
    lab_sorted = lab[order]
    cuts = np.concatenate([[0], np.cumsum(np.bincount(lab_sorted))])
    cc = [np.sort(order[cuts[i]: cuts[i + 1]]) for i in range(nc)]
    return sorted(cc, key=lambda x: -len(x))


def connected_sets_from_pattern(indptr, indices, n, extra_edge=None):
    """`find_connected_sets` on a CSR sparsity pattern (built on the device, `hamsm_pattern_csr`) -- optionally with one more
    edge, inserted in sorted position so that the graph handed to scipy is the one the dense path hands it"""
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csgraph

    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int32)
    if extra_edge is not None:
        r, c = extra_edge
        row = indices[indptr[r]: indptr[r + 1]]
        k = int(np.searchsorted(row, c))
        if k == len(row) or row[k] != c:
            indices = np.insert(indices, int(indptr[r]) + k, c)
            indptr = indptr.copy()
            indptr[r + 1:] += 1
    Cs = sp.csr_matrix((np.ones(len(indices), dtype=np.int8), indices, indptr.astype(np.int32)), shape=(n, n))
    nc, lab = csgraph.connected_components(Cs, directed=True, connection="strong")
    order = np.argsort(lab)
    cuts = np.concatenate([[0], np.cumsum(np.bincount(lab[order]))])
    cc = [np.sort(order[cuts[i]: cuts[i + 1]]) for i in range(nc)]
    return sorted(cc, key=lambda x: -len(x))


class ClusteringMixin:
    n_clusters = None
    clusters = None
    post_cluster_model = None

    # ------------------------------------------------------------------------------------------
    def cluster_coordinates(self, n_clusters, streaming=False, first_cluster_iter=None, use_ray=False,
                            stratified=True, iters_to_use=None, store_validation_model=False, progress_bar=None,
                            **_cluster_args):
        """`_clustering.py:142-196`."""
        self.clustering_method = None
        if not stratified:
            raise NotImplementedError("cluster_aggregated (_clustering.py:197-523) is legacy and out of scope")
        self.clustering_method = "stratified"
        self.cluster_stratified(n_clusters=n_clusters, streaming=streaming, first_cluster_iter=first_cluster_iter,
                                use_ray=use_ray, iters_to_use=iters_to_use, **_cluster_args)
        if store_validation_model:
            self.post_cluster_model = deepcopy(self)

    # ------------------------------------------------------------------------------------------
    def _configure_device(self, bin_mapper, D):
        """Push geometry / mapper / bounds (the state `predict` reads, hamsm.h 'model state').  Dispatch is by class name, as
        in the reference (`SUPPORTED_MAPPERS`, `_clustering.py:598-609`): rectilinear and Euclidean Voronoi mappers are
        evaluated on the device; a Voronoi mapper with a user distance function keeps its own `assign` (host) and the
        library reads the bins from arrays (`_frame_bins`)."""
        h = self.h
        name = type(bin_mapper).__name__
        h.set_geometry(D, self.pcoord_ndim, bin_mapper.nbins)
        self._mapper_external = False
        if name == "RectilinearBinMapper":
            h.set_mapper_rectilinear([np.asarray(b, dtype=np.float32) for b in bin_mapper.boundaries])
        elif name == "VoronoiBinMapper":
            if getattr(bin_mapper.dfunc, "hamsm_kind", None) == "euclidean" and not getattr(bin_mapper, "dfargs", None) \
                    and not getattr(bin_mapper, "dfkwargs", None):
                h.set_mapper_voronoi(np.asarray(bin_mapper.centers, dtype=np.float32).reshape(bin_mapper.nbins, -1))
            else:
                h.set_mapper_external()
                self._mapper_external = True
        else:
            # `_clustering.py:598-609`: functional mappers may re-bin on the fly -- unsupported there too
            raise Exception(f"{type(bin_mapper)} mapper loaded, but supported mappers are RectilinearBinMapper and "
                            f"VoronoiBinMapper. Please provide a supported user_bin_mapper.")
        h.set_bounds(self.basis_pcoord_bounds, self.target_pcoord_bounds)
        h._active_clusters = None

    def _frame_bins(self, which, bin_mapper, store=None):
        """external mapper: device int32 bins of every staged frame's parent (`p0`) / child (`p1`) pcoord, computed ONCE by
        the mapper's own `assign` on the host and cached on the store"""
        store = self._store if store is None else store
        cache = store.__dict__.setdefault("_ext_bins", {})
        key = (which, id(bin_mapper))
        if key not in cache:
            p = getattr(store, which).to_host().reshape(store.N, -1)
            cache[key] = (self.h.to_device(np.asarray(bin_mapper.assign(p), dtype=np.int64).astype(np.int32)), bin_mapper)
        return cache[key][0]

    def cluster_stratified(self, n_clusters, streaming=True, first_cluster_iter=None, use_ray=True, bin_iteration=2,
                           iters_to_use=None, user_bin_mapper=None, progress_bar=None, **_cluster_args):
        """`_clustering.py:525-746`."""
        if user_bin_mapper is None:
            raise NotImplementedError("loading the bin mapper from west.h5 (_clustering.py:584-609) needs westpa; "
                                      "pass user_bin_mapper")
        bin_mapper = user_bin_mapper
        store = self.stage_iterations()
        self._configure_device(bin_mapper, store.D)
        h = self.h
        clusters = StratifiedClusters(bin_mapper, self, n_clusters, [], h=h, **_cluster_args)

        S = bin_mapper.nbins
        self.dtrajs = []
        # sharded runs (SURVEY §8e "exact" mode): rank 0 trains -- on every rank's iterations (`train_shards == "all"`: the
        # frames were gathered to it once, DataMixin.training_store) or on its own block -- and the model is broadcast.
        # `train_shards == "strata"`: EVERY rank holds every iteration's training frames and trains the strata s with
        # s % nranks == rank -- the strata's chains of partial_fits are independent (one MiniBatchKMeans per bin,
        # `_clustering.py:890-911`), so the model is the single-process one bit for bit; then stratum s travels from its rank.
        R, rank = getattr(h, "nranks", 1), getattr(h, "rank", 0)
        by_strata = R > 1 and self.train_shards == "strata"
        if by_strata and not all(isinstance(m.random_state, numbers.Integral) for m in clusters.cluster_models):
            # models that share one RandomState object (random_state=None or an instance, sklearn semantics) draw from it in call
            # order across strata: their chains are not independent
            raise ValueError('train_shards = "strata" needs an integer random_state (independent streams per stratum); '
                             'use train_shards = "all"')
        trains_here = by_strata or rank == 0
        plan_given = None
        if by_strata:
            # every rank receives the rows of ITS strata only (one grouped exchange, cached per (clusters, iterations, mapper)),
            # with the global plan the rows were dealt by
            tstore, t_maxIter, plan_given = self._strata_training_store(n_clusters, iters_to_use, first_cluster_iter, bin_mapper)
        else:
            tstore, t_maxIter = self.training_store()
        all_filled = set()
        train_error = None
        if trains_here:
            # with more than one rank NOTHING rank-local may raise between here and the exchange below -- the other
            # ranks would wait inside a collective for ever.  A failure of this rank's training phase (n_samples < n_clusters,
            # a library error, out of memory ...) is carried into `_all_ranks_ok` and raised on EVERY rank.
            try:
                all_filled = self._train_on_this_rank(clusters, tstore, t_maxIter, iters_to_use, first_cluster_iter, n_clusters,
                                                      bin_mapper, by_strata, S, R, rank, plan_given=plan_given)
                # :718-731 never-trained bins are remapped to the nearest trained one
                for ub in np.setdiff1d(range(S), list(all_filled)):
                    clusters.we_remap[int(ub)] = int(self.find_nearest_bin(bin_mapper, ub, list(all_filled)))
            except Exception as e:  # noqa: BLE001
                if R == 1:
                    raise
                train_error = e
        if R > 1:
            self._all_ranks_ok(train_error, "training")
        if getattr(h, "nranks", 1) > 1:
            # exchange step 1: centroids / counts / ||c||^2 of every stratum and the remap table, rank 0 -> all
            # (replaces ray.put(deepcopy(clusters)), _clustering.py:1171-1174)
            if trains_here:
                clusters._activate()
            remap = np.array([clusters.we_remap[b] for b in range(S)], dtype=np.int32)
            d_remap = h.to_device(remap)
            if by_strata:
                h.bcast_strata(np.arange(S) % R)
                # what the streaming bookkeeping of a stratum's model holds on its owner, for every rank's copy (the RandomState
                # streams themselves stay with the owner)
                book = np.zeros((S, 3), dtype=np.int64)
                for b, cm in enumerate(clusters.cluster_models):
                    if b % R == rank and cm.fitted:
                        book[b] = (1, getattr(cm, "n_steps_", 0), getattr(cm, "_n_since_last_reassign", 0))
                d_book = h.to_device(book)
                h.allreduce_sum_i64(d_book, book.size)
                book = d_book.to_host().reshape(S, 3)
                d_book.free()
            else:
                h.bcast_model(0)
            h.bcast(d_remap, remap.nbytes, 0)
            remap = d_remap.to_host()
            d_remap.free()
            if by_strata or not trains_here:
                clusters._adopt_device_state()
                for b in range(S):
                    clusters.we_remap[b] = int(remap[b])
            if by_strata:
                for b, cm in enumerate(clusters.cluster_models):
                    if b % R != rank and book[b, 0]:
                        cm.n_steps_, cm._n_since_last_reassign = int(book[b, 1]), int(book[b, 2])

        self.clusters = clusters
        self.clusters.model = self
        self.n_clusters = n_clusters * S  # :742 (quirk Q3)
        self.clusters.toggle = False
        self.launch_ray_discretization(progress_bar)

    def _train_on_this_rank(self, clusters, tstore, t_maxIter, iters_to_use, first_cluster_iter, n_clusters, bin_mapper,
                            by_strata, S, R, rank, plan_given=None):
        """the rank-local training phase of `cluster_stratified`: plan from the histogram, then the planned (or stepwise) pass over
        the strata this rank fits.  Returns the set of bins some call fits (the same on every rank: a function of the histogram)."""
        h = self.h
        if iters_to_use is None:  # :640-651
            iters_to_use = range(first_cluster_iter if first_cluster_iter is not None else 1, t_maxIter)
        if not (isinstance(iters_to_use, range) and iters_to_use.step == 1):
            iters_to_use = [int(i) for i in iters_to_use]
        # a5 planning support: raw bin (or -1 in basis/target) of every frame's PARENT pcoord
        # (:849-852 pcoord0List, quirk Q2) + per-iteration histograms, one pass over p0
        d_bin = plan_given[2] if plan_given is not None else h.alloc(max(tstore.N, 1), np.int32)
        d_rows = None
        try:
            d_rows = h.alloc(max(tstore.N, 1), np.int32)
            if plan_given is not None:
                # `train_shards = "strata"`: the plan of the WHOLE run (from the all-reduced histogram, the same on every rank) and
                # the bins the received frames are indexed under (`_strata_training_store`); this rank fits the strata it owns
                plan, all_filled = dict(plan_given[0]), set(plan_given[1])
                mine = (np.arange(S) % R) == rank
                plan["fit"] = plan["fit"] & mine[None, :]
            else:
                if self._mapper_external:
                    h.set_frame_bins(self._frame_bins("p0", bin_mapper, tstore))
                hist = h.train_prepass(tstore.p0, tstore.N, tstore.offsets, d_bin)
                plan, all_filled = self._plan_training_calls(hist, iters_to_use, n_clusters, bin_mapper, store=tstore, maxIter=t_maxIter)
            # (the plan's row counts describe the store's own frames unless the store is a rank's share of the strata)
            self._rows_known = plan_given is None
            done = False
            if getattr(self, "speculative_training", True):
                done = self._train_planned(plan, clusters, tstore, d_bin, d_rows)
            if not done:
                self._train_stepwise(plan, clusters, tstore, d_bin, d_rows)
            self.training_path = "planned" if done else "stepwise"
        finally:
            if plan_given is None:
                d_bin.free()
            if d_rows is not None:
                d_rows.free()
        return all_filled

    def _strata_training_store(self, n_clusters, iters_to_use, first_cluster_iter, bin_mapper):
        """`train_shards = "strata"`: (store, maxIter, (plan, all_filled, d_bin)) this rank trains on.  Stratum s trains on rank
        s % R.  Which rows a stratum's call sees is the reference's, quirk included: a training call lists, per (remapped) bin,
        the POSITIONS of its rows among the call's rows whose parent pcoord is outside basis / target, and applies those
        positions to the call's UNFILTERED frames (`_clustering.py:874` vs `:892-896`; `hamsm_train_build_rows`).  So the
        frame a stratum needs is "frame number r of the call", where r is the filtered position of a row whose bin it owns.
        Which calls exist is a function of the per-iteration bin histogram of the WHOLE run (`_plan_training_calls`).  Hence:
        histograms (and iteration sizes) all-reduced, the plan formed on every rank, every rank packs -- per destination -- the
        frames of its own iterations that the destination's strata will index (`hamsm_gather_rows`), each with its weight and
        with the BIN of the row that selects it, and ONE grouped exchange per array (`hamsm_exchange`) moves them.  Ranks hold
        consecutive blocks of iterations, so what arrives in rank order is in call order, positions ascending: the receiver's
        row r of a call IS the frame its r-th row selects -- it trains on exactly the frames, in exactly the order, a single
        process would.  Round 4 sent every frame to every rank (R rooted gathers; 12.8 GB per rank at C5 whatever R)."""
        h = self.h
        R, rank = h.nranks, h.rank
        S = bin_mapper.nbins
        key = (int(n_clusters), None if iters_to_use is None else tuple(int(i) for i in iters_to_use), first_cluster_iter,
               type(bin_mapper).__name__, tuple(np.asarray(b, dtype=np.float64).tobytes() for b in getattr(bin_mapper, "boundaries", ())),
               np.asarray(getattr(bin_mapper, "centers", ()), dtype=np.float64).tobytes())
        cache = self.__dict__.get("_strata_store")
        if cache is not None and cache[0] == key:
            return cache[1]
        store = self.stage_iterations()
        n_it = max(self.maxIter - 1, 0)                      # local iterations 1..n_it are discretized / trained on
        offs = np.asarray(store.offsets[: n_it + 1], dtype=np.int64)
        N_loc = int(offs[-1]) if n_it else 0
        nan = store.nan_child is not None

        def allsum(a):
            d = h.to_device(np.ascontiguousarray(a, dtype=np.int64))
            h.allreduce_sum_i64(d, a.size)
            out = d.to_host().reshape(a.shape)
            d.free()
            return out

        # (a) how many iterations every rank holds, frame layout agreed
        head = np.zeros(3 * R, dtype=np.int64)
        head[rank], head[R + rank], head[2 * R + rank] = n_it, store.x_dtype * 4096 + store.D, int(nan)
        head = allsum(head)
        cnt = head[:R]
        if head[2 * R:].any():
            raise NotImplementedError("NaN-coordinate segments in a sharded run with train_shards='strata'")
        assert (head[R: 2 * R] == store.x_dtype * 4096 + store.D).all(), "ranks staged different frame layouts"
        T, lo = int(cnt.sum()), int(cnt[:rank].sum())
        first_it = np.concatenate([[0], np.cumsum(cnt)])       # rank r holds global iterations first_it[r] + 1 .. first_it[r + 1]
        # (b) raw bin (-1: basis / target) of every local row's parent pcoord; per-iteration histograms and sizes, all-reduced
        d_bin = h.alloc(max(N_loc, 1), np.int32)
        tab = np.zeros((T + 1, S + 1), dtype=np.int64)         # (+ one empty row: `iters_to_use` may name iteration T + 1)
        bins = np.zeros(0, dtype=np.int64)
        if N_loc:
            if self._mapper_external:
                h.set_frame_bins(self._frame_bins("p0", bin_mapper, store))
            tab[lo: lo + n_it, :S] = h.train_prepass(store.p0, N_loc, offs, d_bin)
            tab[lo: lo + n_it, S] = np.diff(offs)
            bins = d_bin.to_host()[:N_loc].astype(np.int64)
        d_bin.free()
        tab = allsum(tab)
        hist_g, sizes_g = tab[:, :S], tab[:T, S]
        # (c) the plan of the whole run, in global iteration numbers 1..T
        it_use = iters_to_use
        if it_use is None:
            it_use = range(first_cluster_iter if first_cluster_iter is not None else 1, T + 1)
        if not (isinstance(it_use, range) and it_use.step == 1):
            it_use = [int(i) for i in it_use]
        plan, all_filled = self._plan_training_calls(hist_g, it_use, n_clusters, bin_mapper, store=None, maxIter=T + 1)
        G = len(plan["first"])
        # per (call, raw bin): the rank whose stratum the bin's rows feed, or -1 (bin not fitted by the call)
        tgt = plan["call_remap"].astype(np.int64) if G else np.zeros((0, S), dtype=np.int64)
        owner = np.where(np.take_along_axis(plan["fit"], tgt, axis=1), tgt % R, -1) if G else np.zeros((0, S), dtype=np.int64)
        ca, cb = plan["first"].astype(np.int64) - 1, plan["last"].astype(np.int64)       # call g = global iterations [ca, cb) (0-based)
        # (d) calls that straddle a shard boundary: their rows' bins are needed by every rank that holds a part of them
        strad = [g for g in range(G) if np.searchsorted(first_it, ca[g], side="right") != np.searchsorted(first_it, cb[g] - 1, side="right")]
        x_its = np.array(sorted({t for g in strad for t in range(int(ca[g]), int(cb[g]))}), dtype=np.int64)
        x_off = np.concatenate([[0], np.cumsum(sizes_g[x_its])]).astype(np.int64) if len(x_its) else np.zeros(1, dtype=np.int64)
        xb = np.zeros(int(x_off[-1]), dtype=np.int64)
        for k, t in enumerate(x_its):
            if lo <= t < lo + n_it:
                xb[x_off[k]: x_off[k + 1]] = bins[offs[t - lo]: offs[t - lo + 1]] + 2      # (0 = not mine; + 2: bins >= -1)
        xb = allsum(xb) - 2 if len(xb) else xb
        # (e) destination of every local FRAME and the bin it is indexed under.  Call by call: the bins of the call's rows in
        # order (local, or from (d)), the filtered ones numbered 0, 1, ...; frame number r of the call belongs to row r's stratum
        dest = np.full(N_loc, -1, dtype=np.int64)
        bin_of_frame = np.full(N_loc, -1, dtype=np.int64)
        recv_rows = np.zeros(R, dtype=np.int64)               # rows this rank receives, by sender
        call_rows = np.zeros(G, dtype=np.int64)               # rows of each call that arrive here
        x_pos = {int(t): k for k, t in enumerate(x_its)}
        # (calls inside one shard, vectorised over this rank's rows: position among the filtered rows since the call's start)
        if N_loc:
            call_of_it = np.full(T, -1, dtype=np.int64)
            for g in range(G):
                call_of_it[ca[g]: cb[g]] = g
            inside = np.ones(G, dtype=bool)
            inside[strad] = False
            it_row = np.repeat(np.arange(n_it, dtype=np.int64), np.diff(offs))
            g_row = call_of_it[lo + it_row]
            ok = (bins >= 0) & (g_row >= 0)
            ok[ok] &= inside[g_row[ok]]
            start_frame = offs[np.maximum(ca[np.maximum(g_row, 0)] - lo, 0)]              # first local frame of the row's call
            nf = np.concatenate([[0], np.cumsum(bins >= 0)])
            pos = start_frame + (nf[:-1] - nf[start_frame])                                # the frame row i selects
            own = owner[np.maximum(g_row, 0), np.maximum(bins, 0)]
            sel = ok & (own >= 0)
            dest[pos[sel]] = own[sel]
            bin_of_frame[pos[sel]] = bins[sel]
        for g in range(G):
            mine_b = plan["fit"][g] & ((np.arange(S) % R) == rank)
            call_rows[g] = int(plan["nrows"][g][mine_b].sum())
        for r in range(R):     # senders of the calls inside one shard
            gs = [g for g in range(G) if g not in set(strad) and first_it[r] <= ca[g] < first_it[r + 1]]
            recv_rows[r] += int(call_rows[gs].sum()) if gs else 0
        for g in strad:        # (rare: a call whose iterations lie on two or more ranks)
            seq = np.concatenate([xb[x_off[x_pos[int(t)]]: x_off[x_pos[int(t)] + 1]] for t in range(int(ca[g]), int(cb[g]))])
            nfb = seq[seq >= 0]
            own_r = owner[g][nfb]                                                         # owner of frame number r of the call
            sz = sizes_g[ca[g]: cb[g]]
            frame_it = np.repeat(np.arange(int(ca[g]), int(cb[g])), sz)[: len(nfb)]       # iteration that holds frame number r
            src_rank = np.searchsorted(first_it, frame_it, side="right") - 1
            for r in range(R):
                recv_rows[r] += int(((own_r == rank) & (src_rank == r)).sum())
            here = (src_rank == rank) & (own_r >= 0)
            if here.any():
                base = int(np.concatenate([[0], np.cumsum(sz)])[max(lo - int(ca[g]), 0)])  # call-relative number of my first frame
                loc0 = int(offs[max(int(ca[g]) - lo, 0)])
                r_idx = np.nonzero(here)[0]
                dest[loc0 + r_idx - base] = own_r[r_idx]
                bin_of_frame[loc0 + r_idx - base] = nfb[r_idx]
        if N_loc:
            order = np.argsort(dest, kind="stable")
            idx = order[dest[order] >= 0]
            send_rows = np.bincount(dest[dest >= 0], minlength=R).astype(np.int64)
        else:
            idx, send_rows = np.zeros(0, dtype=np.int64), np.zeros(R, dtype=np.int64)
        Nr = int(recv_rows.sum())
        assert Nr == int(call_rows.sum()), (Nr, int(call_rows.sum()))
        # (f) the receiver's frames lie call by call: all of a call's rows under its first iteration
        per_it = np.zeros(T, dtype=np.int64)
        per_it[ca] = call_rows
        offsets_r = np.concatenate([[0], np.cumsum(per_it)]).astype(np.int64)
        # (g) pack and exchange: child frames (`_data.py:616`), weights, and the bin each frame is indexed under
        esz = 4 if store.x_dtype == _abi.F32 else 8
        d_idx = h.to_device(idx if len(idx) else np.zeros(1, dtype=np.int64))
        d_bsrc = h.to_device(bin_of_frame.astype(np.int32) if N_loc else np.zeros(1, dtype=np.int32))
        got = []
        for src, row_bytes, dt, shape in ((store.Xc, store.D * esz, np.float32 if esz == 4 else np.float64, (max(Nr, 1), store.D)),
                                          (store.w, 8, np.float64, (max(Nr, 1),)), (d_bsrc, 4, np.int32, (max(Nr, 1),))):
            d_send = h.alloc(max(len(idx), 1) * row_bytes, np.uint8)
            h.gather_rows(src, row_bytes, d_idx, len(idx), d_send)
            d_recv = h.alloc(shape, dt)
            h.exchange(d_send, send_rows * row_bytes, d_recv, recv_rows * row_bytes)
            h.sync()
            d_send.free()
            got.append(d_recv)
        d_idx.free()
        d_bsrc.free()
        ts = DeviceStore(h, store.D, store.ndim, store.x_dtype, offsets_r, None, got[0], None, None, got[1])
        out = (ts, T + 1, (plan, all_filled, got[2]))
        old = self.__dict__.get("_strata_store")
        if old is not None:
            for buf in (old[1][0].Xc, old[1][0].w, old[1][2][2]):
                buf.free()
        self._strata_store = (key, out)
        self.strata_exchange_stats = {"rows_sent": int(send_rows.sum()), "rows_received": Nr, "rows_local": N_loc,
                                      "rows_all_ranks": int(sizes_g.sum()), "bytes_received": Nr * (store.D * esz + 12),
                                      "calls_across_shards": len(strad)}
        return out

    def _all_ranks_ok(self, error, what):
        """a collective success flag (min over the ranks of `error is None`): every rank leaves with the same outcome -- the
        rank that failed re-raises its own exception, the others raise a RuntimeError naming the phase"""
        h = self.h
        d = h.to_device(np.array([0.0 if error is not None else 1.0]))
        h.allreduce_min_f64(d, 1)
        ok = float(d.to_host()[0]) > 0.5
        d.free()
        if error is not None:
            raise error
        if not ok:
            raise RuntimeError(f"{what} failed on another rank (its own exception is raised there); this rank stops with it "
                               "instead of waiting in the next collective")

    # ------------------------------------------------------------------------------------------
    def _plan_training_calls(self, hist, iters_to_use, n_clusters, bin_mapper, store=None, maxIter=None):
        """The control flow of `cluster_stratified` / `do_stratified_clustering` (`_clustering.py:664-716,
        794-886`): which iterations each training call consumes and which bins it fits.  It depends only
        on the per-iteration bin histogram, so every call is known before any arithmetic runs.

        Returns (plan, all_filled); plan = arrays over calls: first / last iteration, call_remap [G,S],
        nrows [G,S] (rows per bin after the call's remap) and fit [G,S] (bins fitted by the call)."""
        S = bin_mapper.nbins
        maxIter = self.maxIter if maxIter is None else maxIter
        its = (np.arange(iters_to_use.start, iters_to_use.stop, dtype=np.int64) if isinstance(iters_to_use, range)
               else np.asarray(iters_to_use, dtype=np.int64))
        n = len(its)
        ident = np.arange(S, dtype=np.int32)
        valid = its <= maxIter
        H = np.zeros((n, S), dtype=np.int64)
        H[valid] = hist[its[valid] - 1]
        # iterations with NaN child coordinates: `get_iter_coordinates` drops those rows (_data.py:549-553), the call's
        # `assert pcoord_array.shape[0] == iter_coords.shape[0]` (_clustering.py:860-862) then fails -- fatal on the first
        # call, swallowed afterwards (:693-705), i.e. the call is lost and the loop advances by one iteration
        nan_it = np.zeros(maxIter + 2, dtype=bool)
        if store is not None and getattr(store, "nan_child", None) is not None:
            per_it = np.add.reduceat(store.nan_child.astype(np.int64), store.offsets[:-1][np.diff(store.offsets) > 0])
            nan_it[np.nonzero(np.diff(store.offsets) > 0)[0] + 1] = per_it > 0
        # an iteration that fills every bin it touches is a training call by itself (:885-886)
        alone = valid & np.all((H >= n_clusters) | (H == 0), axis=1) & ~nan_it[np.minimum(its, maxIter + 1)]
        stops = np.nonzero(~alone)[0]
        firsts, lasts, remaps, nrows, fits = [], [], [], [], []
        pos, first_call = 0, True
        while pos < n:
            if alone[pos]:
                k = np.searchsorted(stops, pos)
                end = int(stops[k]) if k < len(stops) else n
                sl = slice(pos, end)
                fit = H[sl] > 0
                keep = fit.any(axis=1)
                firsts.append(its[sl][keep])
                lasts.append(its[sl][keep])
                remaps.append(np.broadcast_to(ident, (int(keep.sum()), S)))
                nrows.append(H[sl][keep])
                fits.append(fit[keep])
                pos, first_call = end, False
                continue
            # pull iterations until every SEEN bin has >= k rows
            group, counts = [], np.zeros(S, dtype=np.int64)
            call_remap = ident.copy()
            ran_out = False
            nan_failed = False
            while True:
                if pos + len(group) >= n:  # :798-826 IndexError branch
                    ran_out = True
                    break
                it = int(its[pos + len(group)])
                if it > maxIter:  # :834-840
                    break
                group.append(it)
                if nan_it[it]:
                    if first_call:
                        raise AssertionError(f"iteration {it}: coordinates of {int(hist[it - 1].sum())}+ segments vs NaN-free "
                                             "coordinates differ in length (NaN coordinates in the first clustering call)")
                    log.info("Clustering couldn't use last iteration, not all bins filled.")
                    nan_failed = True
                    break
                counts += hist[it - 1]
                seen = counts > 0
                if np.all(counts[seen] >= n_clusters):  # :885-886
                    break
            if not all(b == a + 1 for a, b in zip(group, group[1:])):
                raise NotImplementedError("non-contiguous iters_to_use within one training call")
            unique_bins = np.nonzero(counts > 0)[0]
            failed = nan_failed
            if ran_out and not failed:
                unfilled = unique_bins[counts[unique_bins] < n_clusters]
                filled = np.setdiff1d(unique_bins, unfilled)
                try:
                    for ub in unfilled:
                        call_remap[ub] = self.find_nearest_bin(bin_mapper, ub, list(filled))  # :811-821
                except AssertionError:
                    # :693-705: fatal on the very first call, swallowed afterwards
                    if first_call:
                        raise
                    log.info("Clustering couldn't use last iteration, not all bins filled.")
                    failed = True
                unique_bins = filled
            if not failed and len(group) and len(unique_bins):
                nr = np.zeros(S, dtype=np.int64)
                np.add.at(nr, call_remap, counts)
                fit = np.zeros(S, dtype=bool)
                fit[unique_bins] = True
                firsts.append(np.array([group[0]]))
                lasts.append(np.array([group[-1]]))
                remaps.append(call_remap[None, :])
                nrows.append(nr[None, :])
                fits.append(fit[None, :])
            first_call = False
            # :664-669,716: a call that consumed extra iterations skips them; a failed call only advances by one
            pos += max(len(group), 1) if not failed else 1
        if firsts:
            plan = dict(first=np.concatenate(firsts), last=np.concatenate(lasts),
                        call_remap=np.ascontiguousarray(np.concatenate(remaps), dtype=np.int32),
                        nrows=np.concatenate(nrows), fit=np.concatenate(fits))
        else:
            plan = dict(first=np.zeros(0, np.int64), last=np.zeros(0, np.int64), call_remap=np.zeros((0, S), np.int32),
                        nrows=np.zeros((0, S), np.int64), fit=np.zeros((0, S), bool))
        all_filled = set(np.nonzero(plan["fit"].any(axis=0))[0].tolist())
        return plan, all_filled

    def _train_stepwise(self, plan, clusters, store, d_bin, d_rows):
        """One `partial_fit` at a time with the host in the loop (RandomState draws for reassignment)."""
        h = self.h
        for g in range(len(plan["first"])):
            f0, f1 = store.span(int(plan["first"][g]), int(plan["last"][g]))
            n = f1 - f0
            cnt, starts = h.train_build_rows(d_bin.offset(f0), n, plan["call_remap"][g], d_rows)
            X_base = store.Xc.offset(f0 * store.D)  # segment-END frames (_data.py:616)
            w_base = store.w.offset(f0) if self.use_weights_in_clustering else None
            w_host_all = rows_host = None
            if self.use_weights_in_clustering:
                w_host_all = np.empty(n)
                h.d2h(w_host_all, store.w.offset(f0))
            for b in np.nonzero(plan["fit"][g])[0]:  # :890 ascending (np.unique)
                nb = int(cnt[b])
                rows_b = d_rows.offset(int(starts[b]))
                w_rows = None
                if w_host_all is not None:
                    if rows_host is None:
                        rows_host = np.empty(int(cnt.sum()), dtype=np.int32)
                        h.d2h(rows_host, d_rows, rows_host.nbytes)
                    w_rows = w_host_all[rows_host[int(starts[b]): int(starts[b]) + nb]]
                clusters.cluster_models[int(b)].partial_fit_device(X_base, store.x_dtype, store.D, rows_b, nb, w_base, w_rows)

    def _train_planned(self, plan, clusters, store, d_bin, d_rows):
        """Execute the whole plan on the device without per-step host round trips.

        sklearn's step needs the host only when a low-count centroid must be re-seeded from a random
        batch row (`_kmeans.py:1643-1673`).  The plan *predicts* that this never happens (it does not
        on data whose strata hold at least as many modes as centroids), enqueues every seeding and every
        step, and checks the prediction afterwards with what the device reported per step (was any count
        zero before the step; how many counts fell under ratio*max after it).  A wrong prediction costs
        a rerun through `_train_stepwise` from restored RandomState streams -- results are identical
        either way.  Returns False when the rerun is needed."""
        from .stratified_clustering import _check_random_state

        h = self.h
        G = len(plan["first"])
        if G == 0 or not plan["fit"].any():
            # nothing to fit here: no call at all, or (train_shards = "strata") a rank that owns no fitted stratum -- more ranks
            # than strata, or every stratum s with s % nranks == rank never filled (this used to die on `z[0]`
            # below while the other ranks were already inside hamsm_bcast_strata)
            return True
        models = clusters.cluster_models
        ratios = {m.reassignment_ratio for m in models}
        if len(ratios) != 1:
            return False
        ratio = ratios.pop()
        offs = store.offsets
        gb = offs[plan["first"] - 1]
        ge = offs[plan["last"]]
        fit = plan["fit"]
        if getattr(self, "_rows_known", False) and os.environ.get("HAMSM_CHECK_ROWS", "0") != "1":
            # the row lists' sizes ARE the prepass histogram pushed through the calls' remaps (`plan["nrows"]`, all strata) and
            # their starts the running sums over the strata behind the group's first frame (`hamsm_train_build_rows`): nothing
            # to read back, the kernel is enqueued and the seedings follow at once (round 5: the read-back was a synchronisation
            # point of ~0.3 ms in front of every pass; HAMSM_CHECK_ROWS=1 reads back and compares)
            h.train_build_rows_batched(d_bin, gb, ge, plan["call_remap"], d_rows, readback=False)
            counts = np.ascontiguousarray(plan["nrows"], dtype=np.int64)
            starts = np.asarray(gb, dtype=np.int64)[:, None] + np.cumsum(counts, axis=1) - counts
        else:
            counts, starts = h.train_build_rows_batched(d_bin, gb, ge, plan["call_remap"], d_rows)
            assert np.array_equal(counts[fit], plan["nrows"][fit]), "row lists disagree with the prepass histogram"
            if getattr(self, "_rows_known", False):
                kn = np.ascontiguousarray(plan["nrows"], dtype=np.int64)
                assert np.array_equal(counts, kn) and np.array_equal(starts, np.asarray(gb, dtype=np.int64)[:, None] + np.cumsum(kn, axis=1) - kn)
        use_w = self.use_weights_in_clustering
        d_w = store.w if use_w else None
        # RandomState streams as they were, for the rerun
        # (a stream created here from an integer seed is restored by seeding it again: no copy of its state up front)
        rs_objs, saved, fresh = {}, {}, {}
        for m in models:
            if "_random_state" not in m.__dict__:
                m._random_state = _check_random_state(m.random_state, m)
                if isinstance(m.random_state, numbers.Integral):
                    fresh[id(m._random_state)] = m.random_state
            rs_objs[id(m._random_state)] = m._random_state
        for k, r in rs_objs.items():
            if k not in fresh:
                saved[k] = r.get_state()
        clusters._activate()
        # per bin: the calls that fit it, in order; bookkeeping of MiniBatchKMeans predicted in closed form
        # (_kmeans.py:2029-2043: with no empty centroid, do_reassign fires every ceil(10K / batch_size) steps)
        # (only the scalars the seedings need are formed here: the per-step arrays of each stratum are built below, while the
        # device is already seeding)
        pred, parts = {}, []
        first_g = np.full(len(models), G, dtype=np.int64)
        for b in np.nonzero(fit.any(axis=0))[0]:
            b = int(b)
            m = models[b]
            if m.fitted:
                return False  # continuing a fitted model: take the stepwise path
            gs = np.nonzero(fit[:, b])[0]
            nb0 = int(counts[gs[0], b])
            K = m.n_clusters
            if nb0 < K:
                raise ValueError(f"n_samples={nb0} should be >= n_clusters={K}.")  # _kmeans.py:875-879
            bs = int(min(m.batch_size, nb0))
            init_size = m.init_size
            if init_size is None:
                init_size = 3 * bs
                if init_size < K:
                    init_size = 3 * K
            elif init_size < K:
                init_size = 3 * K
            init_size = int(min(init_size, nb0))
            period = max(1, -(-10 * K // bs))
            pred[b] = dict(bs=bs, init=init_size, n_since=int((len(gs) - 1) % period) * bs, nsteps=len(gs), gs=gs, period=period)
            first_g[b] = gs[0]
        # seeding draws in the stepwise order: by first call, then bin (shared RandomState streams stay in sync)
        seeds = []
        for b in sorted(pred, key=lambda x: (first_g[x], x)):
            m, g0 = models[b], int(first_g[b])
            nb, K, rs = int(counts[g0, b]), m.n_clusters, m._random_state
            w_host = np.ones(nb)
            if use_w:
                rows_host = np.empty(nb, dtype=np.int32)
                h.d2h(rows_host, d_rows.offset(int(starts[g0, b])), rows_host.nbytes)
                w_grp = np.empty(int(ge[g0] - gb[g0]))
                h.d2h(w_grp, store.w.offset(int(gb[g0])))
                w_host = w_grp[rows_host]
            sub, n = None, nb
            if pred[b]["init"] < nb:  # _kmeans.py:1003-1011
                sub = rs.randint(0, nb, pred[b]["init"])
                n = pred[b]["init"]
                w_host = w_host[sub]
            fidx = rs.choice(n, p=w_host / w_host.sum())  # :225
            L = 2 + int(np.log(K))
            u = rs.uniform(size=(K - 1) * L)  # :243, K-1 calls of uniform(size=L): the same stream as one call
            seeds.append((b, int(gb[g0]), int(starts[g0, b]), nb, sub, K, int(fidx), u, L))
        # strata are independent and a seeding reads only its own first batch: every seeding can be issued
        # before every step
        z = list(zip(*seeds))
        h.kmeans_seed_multi(z[0], store.Xc, store.x_dtype, store.D, z[1], d_rows, z[2], z[3], d_w, z[4], z[5], z[6], z[7], z[8])
        for b, pb in pred.items():  # the steps of every stratum, call by call
            gs, period = pb.pop("gs"), pb.pop("period")
            t = np.arange(len(gs))
            do = (t % period) == 0
            st = np.zeros(len(gs), dtype=h.STEP_DTYPE)
            st["stratum"], st["flags"] = b, (do & (ratio > 0)).astype(np.int32)
            st["x_off"], st["rows_off"], st["n_b"] = gb[gs], starts[gs, b], counts[gs, b]
            st["n_span"] = np.minimum(ge[gs] - gb[gs], 2 ** 31 - 1)   # row indices of a call range over its group's frames
            parts.append((gs, np.full(len(gs), b), st, t == 0))

        def undo():
            for k, r in rs_objs.items():
                if k in fresh:
                    r.seed(fresh[k])
                else:
                    r.set_state(saved[k])
            for b in pred:
                h.set_centroids(b, None)
                m = models[b]
                m.__dict__["_device_newer"] = False
                m.__dict__["_host_centers"] = None
                m.__dict__.pop("n_steps_", None)
            return False

        shared_rng = len(rs_objs) < len(pred)  # models share a RandomState: draws interleave in call order
        # Everything is fused, first steps included (flag FIRST: their counts are zero by construction, which is what
        # makes the low-count test fire on them; with a seeding subsample drawn with replacement, duplicate seeds end
        # that step with no member and are re-seeded at once -- the usual stop / settle / resume).  The low-count rule
        # (`_kmeans.py:1643-1673`) consumes `random_state.permutation(n_b)` only when some count IS low, so the host predicts per stratum whether it
        # will be ("as at the last test": the outcome is persistent -- never on balanced data, at every test once a
        # stratum holds more centroids than populated modes) and hands the permutation heads over; the device
        # re-seeds from them and stops a stratum's chain at the first wrong prediction, which is then settled here
        # (RandomState rewound or the re-seeding done on the host) before the chain resumes.
        chain = {}
        for gs, bb, st, first in parts:
            st = st.copy()
            st["flags"] |= np.where(first, 2, 0).astype(np.int32)
            chain[int(bb[0])] = dict(g=gs, st=st, pos=0)
            models[int(bb[0])].__dict__["_device_newer"] = True
        predict = {b: False for b in chain}
        # tests a stratum may run ahead on a "low counts" prediction: one at first (a speculative permutation costs 0.15 ms of host
        # time at 12 500-row batches and the first prediction after two low tests is the likeliest to fail -- C5: all four strata,
        # 16 permutations drawn for nothing when this started at 4: 127.8 -> 123.3 ms per step), doubling while it keeps holding
        PT0 = 1
        pred_tests = {b: PT0 for b in chain}
        low_streak = {b: 0 for b in chain}   # consecutive tests of a stratum that found low counts and stopped its chain
        total_calls = int(max((len(c["g"]) for c in chain.values()), default=0))
        span = G  # calls per fused run; 64 after a stop, four times as many after every clean run
        t_prev = time.perf_counter()
        g_lo = 0
        n_runs = 0
        while any(c["pos"] < len(c["g"]) for c in chain.values()):
            live = [b for b, c in chain.items() if c["pos"] < len(c["g"])]
            # every live stratum contributes its NEXT `span` steps, whatever their call numbers, and the run interleaves them
            # position by position: a launch group of the device (a maximal run of distinct strata) then always carries every
            # live stratum.  Strata are independent chains (own model, own RandomState; shared ones run stepwise), so the
            # order between strata is free -- cutting runs by call RANGE instead made a stratum that had stopped early catch
            # up alone, a quarter of the chip at a time (C5: 2.5 launches per call and queue instead of 1).
            sel = {b: (chain[b]["pos"], min(chain[b]["pos"] + span, len(chain[b]["g"]))) for b in live}
            # a stratum that runs on the prediction "low counts again" carries a speculative permutation on EVERY test and stops at
            # the first one that does not need it: it goes only a few tests ahead (1, doubling while the prediction keeps holding)
            # -- a 64-call run of a stratum that tests at every call used to draw 117 permutations to apply 9 steps
            if not shared_rng:
                for b in live:
                    if predict[b]:
                        a, e = sel[b]
                        t = np.flatnonzero(chain[b]["st"]["flags"][a:e] & 1)
                        if len(t) > pred_tests[b]:
                            sel[b] = (a, a + int(t[pred_tests[b] - 1]) + 1)
            gcat = np.concatenate([chain[b]["g"][a:e] for b, (a, e) in sel.items()])
            g_lo, g_hi = int(gcat.min()), int(gcat.max()) + 1
            bcat = np.concatenate([np.full(e - a, b) for b, (a, e) in sel.items()])
            kcat = np.concatenate([np.arange(e - a) for b, (a, e) in sel.items()])
            scat = np.concatenate([chain[b]["st"][a:e] for b, (a, e) in sel.items()])
            order = np.lexsort((bcat, kcat))  # each stratum's steps in its own call order (:890), strata side by side
            steps = scat[order].copy()
            steps["perm_off"], steps["n_perm"] = -1, 0
            # speculative draws, per model in step order (models own their RandomState here; shared ones never predict)
            perms, off, drawn = [], 0, {b: [] for b in sel}
            # (only the tests of strata whose last test found low counts carry a permutation: usually none)
            want = [b for b in sel if predict[b]] if not shared_rng else []
            tested = np.nonzero((steps["flags"] & 1).astype(bool) & np.isin(steps["stratum"], want))[0] if want else ()
            for i in tested:
                b = int(steps["stratum"][i])
                m = models[b]
                nb_i = int(steps["n_b"][i])
                drawn[b].append((int(i), m._random_state.get_state()))
                head = m._random_state.permutation(nb_i)[: min(m.n_clusters, nb_i)]
                steps["perm_off"][i], steps["n_perm"][i] = off, len(head)
                perms.append(head)
                off += len(head)
            t_run = time.perf_counter()
            report, resume, reason = h.minibatch_run(store.Xc, store.x_dtype, store.D, d_rows, d_w, steps, ratio,
                                                     np.concatenate(perms) if perms else None)
            n_runs += 1
            if os.environ.get("HAMSM_DEBUG_REPORT"):
                t_now = time.perf_counter()
                applied = {b: int(np.searchsorted(np.nonzero(steps["stratum"] == b)[0], resume[b], side="left")) for b in sel}
                print(f"[plan] run {n_runs}: calls {g_lo}..{g_hi}, {len(steps)} steps ({sum(applied.values())} applied), "
                      f"{int((steps['flags'] & 1).sum())} tests ({len(perms)} with a permutation), stops "
                      f"{dict((b, int(reason[b])) for b in sel if reason[b])}; host {1e3 * (t_run - t_prev):.2f} ms before, "
                      f"run {1e3 * (t_now - t_run):.2f} ms", flush=True, file=sys.stderr)
                t_prev = t_now
            stopped = False
            for b, (a, e) in sel.items():
                m = models[b]
                mine = np.nonzero(steps["stratum"] == b)[0]
                n_applied = int(np.searchsorted(mine, resume[b], side="left"))
                why = int(reason[b])
                last = int(mine[n_applied - 1]) if n_applied else -1  # the last applied step of this stratum
                # permutations drawn for steps that were not applied (and, when its prediction failed, for the last
                # applied one) were never consumed by the reference: rewind to the earliest of them
                undo_from = last if why in (h.STOP_UNUSED, h.STOP_CAP) else last + 1
                for i, state in drawn[b]:
                    if i >= undo_from:
                        m._random_state.set_state(state)
                        break
                chain[b]["pos"] = a + n_applied
                if why == h.STOP_NONE:
                    if predict[b]:
                        pred_tests[b] = min(2 * pred_tests[b], 1 << 20)
                    elif n_applied and (steps["flags"][mine[:n_applied]] & 1).any():
                        low_streak[b] = 0   # a test passed without low counts
                    continue
                stopped = True
                pred_tests[b] = PT0
                if why == h.STOP_ZERO or shared_rng or last < 0:
                    log.info("training plan of stratum %d does not hold (stop %d): rerunning stepwise", b, why)
                    return undo()
                if why == h.STOP_UNUSED:
                    predict[b] = False
                    low_streak[b] = 0
                else:  # the step is applied, its re-seeding is still due: on the host, with the model's RandomState
                    st_l = steps[last]
                    m.__dict__["_device_newer"] = True
                    m._host_reassign(store.Xc.offset(int(st_l["x_off"]) * store.D), store.x_dtype, store.D,
                                     d_rows.offset(int(st_l["rows_off"])), int(st_l["n_b"]))
                    m._counts_cache = None
                    # "low again at the next test" is predicted only from the second low test in a row on: right after a re-seeding
                    # the re-seeded clusters carry the smallest kept count, so an isolated low test is usually followed by a
                    # clean one (C5's shape: every low stop was followed by an unused-permutation stop, two runs for one event);
                    # streaks -- a stratum that is low test after test -- switch to the device's own re-seeding after two stops
                    low_streak[b] += 1
                    predict[b] = low_streak[b] >= 2
            # after a stop the next run is short: a stratum that has just been re-seeded carries a speculative permutation on
            # every test (0.1 ms each at 12 500-row batches) and is the likeliest to stop again at its first one
            span = min(64, G) if stopped else min(span * 4, G)
            if n_runs > 64 + total_calls // 8:
                log.info("training plan keeps mispredicting: rerunning stepwise")
                return undo()
        for b, st in pred.items():
            m = models[b]
            m._batch_size, m._init_size = st["bs"], st["init"]
            m._n_since_last_reassign = st["n_since"]
            m._counts_cache = None
            m.n_steps_ = st["nsteps"]  # (the model was unfitted before this pass)
            m.__dict__["_device_newer"] = True
        return True

    # ------------------------------------------------------------------------------------------
    def launch_ray_discretization(self, progress_bar=None):
        """`_clustering.py:1144-1329`: assign the parent and the child frame of every segment of
        iterations 1..maxIter-1.  One fused launch per side instead of a Ray task per iteration."""
        store = self.stage_iterations()
        h = self.h
        self.clusters._activate()
        n_it = self.maxIter - 1
        nseg = int(store.offsets[n_it]) if n_it > 0 else 0
        d_parent = h.alloc(max(nseg, 1), np.int32)
        d_child = h.alloc(max(nseg, 1), np.int32)
        d_seen = h.zeros((2, self.clusters.bin_mapper.nbins), np.int32)
        ext = getattr(self, "_mapper_external", False)
        if nseg and ext:
            h.set_frame_bins(self._frame_bins("p0", self.clusters.bin_mapper))
            h.assign(store.Xp, store.x_dtype, store.D, store.p0, nseg, d_parent, d_seen)  # :1299-1301
            h.set_frame_bins(self._frame_bins("p1", self.clusters.bin_mapper))
            h.assign(store.Xc, store.x_dtype, store.D, store.p1, nseg, d_child, d_seen)   # :1315-1316
        elif nseg:
            # both sides in one call: enqueued back to back, one error read-back
            h.assign_pair(store.Xp, store.Xc, store.x_dtype, store.D, store.p0, store.p1, nseg, d_parent, d_child, d_seen)
        seen = d_seen.to_host()
        d_seen.free()
        if getattr(h, "nranks", 1) > 1:  # which bins held target / basis frames: union over the shards
            d_any = h.to_device(seen.astype(np.int64).ravel())
            h.allreduce_sum_i64(d_any, seen.size)
            seen = d_any.to_host().reshape(seen.shape)
            d_any.free()
        self.clusters.target_bins.update(np.nonzero(seen[0])[0].tolist())  # :1223-1224
        self.clusters.basis_bins.update(np.nonzero(seen[1])[0].tolist())
        self._d_parent, self._d_child, self._n_labelled = d_parent, d_child, nseg
        self._dtrajs_cache = None
        self._pairs_cache = None
        if getattr(store, "nan_child", None) is not None and nseg:
            # the reference discretizes every segment; a NaN frame survives only where the label needs no coordinates
            # (basis / target, stratified_clustering.py:159-169) -- anywhere else sklearn's input validation raises
            Kfit = sum(self.clusters.n_fitted())
            for d_lab, mask in ((d_parent, store.nan_parent[:nseg]), (d_child, store.nan_child[:nseg])):
                if mask.any():
                    lab = np.empty(nseg, dtype=np.int32)
                    h.d2h(lab, d_lab, lab.nbytes)
                    if np.any(lab[mask] < Kfit):
                        raise ValueError("Input contains NaN.")

    # dtrajs / pair_dtrajs are host views of the device labels, materialised on first access -- each by itself: `dtrajs` costs
    # one 8 N-byte device-to-host copy into a page-locked block, `pair_dtrajs` one of 16 N bytes (the
    # pageable copy of both took 13 ms at C3; `_abi.HostBlock`)
    def _labels_host(self, which):
        if getattr(self, "_d_child", None) is None:  # unpickled / never discretized: the plain host lists
            return self.__dict__.get("_dtrajs_plain" if which == 0 else "_pairs_plain", [])
        key = "_dtrajs_cache" if which == 0 else "_pairs_cache"
        if getattr(self, key, None) is None:
            # widened to int64 (and interleaved) on the device; the per-iteration arrays are views of the one host block
            flat = self.h.labels_to_host(self._d_parent, self._d_child, self._n_labelled, child=which == 0, pairs=which == 1)[which]
            offs = self._store.offsets[: self.maxIter].tolist()
            setattr(self, key, [flat[a:b] for a, b in zip(offs[:-1], offs[1:])])
        return getattr(self, key)

    def _labels_device(self):
        """(d_parent, d_child) -- uploaded again from the host lists when the model came out of a pickle"""
        if getattr(self, "_d_child", None) is None:
            pairs = self.__dict__.get("_pairs_plain") or []
            if not len(pairs):
                raise RuntimeError("no discretization: call cluster_coordinates / launch_ray_discretization first")
            flat = np.concatenate([np.asarray(p, dtype=np.int64).reshape(-1, 2) for p in pairs])
            self._d_parent = self.h.to_device(np.ascontiguousarray(flat[:, 0], dtype=np.int32))
            self._d_child = self.h.to_device(np.ascontiguousarray(flat[:, 1], dtype=np.int32))
            self._n_labelled = len(flat)
            self._dtrajs_cache = self._pairs_cache = None
        return self._d_parent, self._d_child

    @property
    def dtrajs(self):
        """list over iterations 1..maxIter-1 of child-frame labels (`_clustering.py:1226,1238`)."""
        return self._labels_host(0)

    @dtrajs.setter
    def dtrajs(self, v):
        self.__dict__["_dtrajs_plain"] = v

    @property
    def pair_dtrajs(self):
        """per iteration an (nSeg, 2) array of (parent, child) labels -- `np.array(pair_dtrajs[i])` of the
        reference's list of tuples (`_clustering.py:1228,1240`; `_fluxmatrix.py:288`)."""
        return self._labels_host(1)

    @pair_dtrajs.setter
    def pair_dtrajs(self, v):
        self.__dict__["_pairs_plain"] = v

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def find_nearest_bin(bin_mapper, bin_idx, filled_bins):
        """`_clustering.py:1331-1396`: Voronoi-style nearest *filled* bin by bin-centre distance."""
        assert len(filled_bins) > 0, ("Can't find nearest populated bin -- no WE bins are populated with clusters! "
                                      "Try fewer clusters/bin.")
        name = type(bin_mapper).__name__
        assert name in ("VoronoiBinMapper", "RectilinearBinMapper"), f"{type(bin_mapper)} is unsupported!"
        if name == "VoronoiBinMapper":
            centers, dist = bin_mapper.centers, bin_mapper.dfunc
        else:
            per_dim = [np.asarray(b)[:-1] + (np.asarray(b)[1:] - np.asarray(b)[:-1]) / 2 for b in bin_mapper.boundaries]
            ndim = len(per_dim)
            # same grid ordering as the reference's meshgrid(...).T.squeeze().reshape(-1, ndim)  (:1378-1380)
            centers = np.array(np.meshgrid(*per_dim)).T.squeeze().reshape(-1, ndim)

            def dist(point, others):
                return np.sqrt(np.mean(np.power(point - others, 2), axis=1))

        ignored = np.setdiff1d(range(centers.shape[0]), filled_bins)
        kept = np.delete(centers, ignored, axis=0)
        with np.errstate(invalid="ignore"):
            closest = int(np.argmin(dist(centers[bin_idx], kept)))
        for ig in sorted(ignored):  # :1391-1394 shift past the deleted indices
            if closest >= ig:
                closest += 1
        return closest

    # ------------------------------------------------------------------------------------------
    def get_cluster_centers(self):
        """`_clustering.py:1528-1599`: per-cluster mean / min / max of the child pcoord (dim 0) over
        the discretized frames, then the pcoord sort order.  Segmented reduction on the device."""
        h = self.h
        store = self.stage_iterations()
        K = self.n_clusters
        ns = K + 2
        _, d_child = self._labels_device()
        n = self._n_labelled
        d_sum = h.zeros(ns, np.float64)
        d_min = h.to_device(np.full(ns, np.inf))
        d_max = h.to_device(np.full(ns, -np.inf))
        d_cnt = h.zeros(ns, np.int64)
        h.cluster_stats(d_child, store.p1, n, ns, d_sum, d_min, d_max, d_cnt)
        if getattr(h, "nranks", 1) > 1:
            # exchange step: per-cluster pcoord sums, extrema and counts of the iteration shards
            h.allreduce_sum_f64(d_sum, ns)
            h.allreduce_min_f64(d_min, ns)
            h.allreduce_max_f64(d_max, ns)
            h.allreduce_sum_i64(d_cnt, ns)
        s, mn, mx, cnt = d_sum.to_host(), d_min.to_host(), d_max.to_host(), d_cnt.to_host()
        for b in (d_sum, d_min, d_max, d_cnt):
            b.free()
        centers = np.zeros((ns, self.pcoord_ndim))
        rng = np.zeros((ns, self.pcoord_ndim, 2))
        # :1544-1555 basis/target rows get `self.{basis,target}_bin_center` (singular) == None -> NaN (quirk Q8)
        centers[K + 1] = np.nan if self.target_bin_center is None else self.target_bin_center
        centers[K] = np.nan if self.basis_bin_center is None else self.basis_bin_center
        rng[K + 1] = centers[K + 1, 0]
        rng[K] = centers[K, 0]
        with np.errstate(invalid="ignore", divide="ignore"):
            mean = s[:K] / cnt[:K]
        empty = cnt[:K] == 0
        for cl in np.nonzero(empty)[0]:
            log.warning(f"No trajectories in cluster {cl}! (Target was {K + 1})")  # :1573-1577
        centers[:K] = np.where(empty, np.nan, mean)[:, None]
        rng[:K, :, 0] = np.where(empty, np.nan, mn[:K])[:, None]
        rng[:K, :, 1] = np.where(empty, np.nan, mx[:K])[:, None]
        order = np.argsort(centers[:, 0])  # :1593 NaNs (basis, target, empty clusters) sort last
        self.targetRMSD_centers = centers[order]
        self.targetRMSD_minmax = rng[order]
        self.cluster_pcoord_counts = np.concatenate([cnt[:K], [0, 0]])[order]
        return order

    def update_sorted_cluster_centers(self):
        """`_clustering.py:1601-1611`."""
        bin_centers = self.targetRMSD_centers[:, 0]
        bin_centers[self.indTargets] = self.target_bin_centers
        bin_centers[self.indBasis] = self.basis_bin_centers
        self.all_centers = bin_centers
        self.sorted_centers = np.argsort(bin_centers)

    # ------------------------------------------------------------------------------------------
    removed_clusters = ()

    def update_cluster_structures(self, build_pcoord_cache=False):
        """`_clustering.py:1398-1526`: the structures (segment-end coordinates), weights and provenance of every cluster, for
        iterations 1..maxIter-2.  The reference appends segment by segment; here the segments are grouped by cluster on the
        device (`hamsm_group_by_label`, stable) and each iteration's coordinates are gathered once."""
        assert self.clusters is not None, "Clusters have not been computed!"
        h = self.h
        _, d_child = self._labels_device()
        last = self.maxIter - 2                                   # `range(1, total_num_iterations - 1)` (:1437, 1464)
        offs = self.stage_iterations().offsets
        n = int(offs[last]) if last >= 1 else 0
        ns = int(self.n_clusters) + 2
        order, indptr = h.group_by_label(d_child, n, ns)
        it_of = np.searchsorted(offs, order, side="right")       # 1-based iteration of every grouped segment
        seg_of = order - offs[it_of - 1]
        coords, weights, pcs, west, segind = {}, {}, {}, {}, {}
        with self._source_session():
            for it in range(1, last + 1):
                # (`get_iter_coordinates` drops NaN rows, and the reference indexes what remains by segment number, :1466-1493:
                # kept as it is -- an IndexError, or rows shifted by one, where an iteration has NaN coordinates)
                coords[it] = self.get_iter_coordinates(it)
                weights[it] = self.seg_weights[it]
                pcs[it] = self.pcoord1List
                west[it], segind[it] = self.westList, self.segindList   # provenance: file index / segment id inside that file
        cluster_structures, cluster_structure_weights, structure_iteration_segments = {}, {}, {}
        pcoord_cache = {} if build_pcoord_cache else None
        # keys appear in the order in which the reference first meets each cluster (dict insertion order)
        first_seen = {c: int(order[indptr[c]]) for c in range(ns) if indptr[c + 1] > indptr[c]}
        for c in sorted(first_seen, key=first_seen.get):
            sl = slice(int(indptr[c]), int(indptr[c + 1]))
            its, segs = it_of[sl], seg_of[sl]
            if c in self.removed_clusters:
                raise Exception("This dtraj point was in a removed cluster -- this should never happen!")
            cluster_structures[c] = [coords[int(i)][int(s)] for i, s in zip(its, segs)]
            cluster_structure_weights[c] = [weights[int(i)][int(s)] for i, s in zip(its, segs)]
            # `[iteration, segindList[seg], fileList[westList[seg]]]` (:1499-1503)
            structure_iteration_segments[c] = [[int(i), int(segind[int(i)][int(s)]), self.fileList[int(west[int(i)][int(s)])]]
                                               for i, s in zip(its, segs)]
            if build_pcoord_cache:
                pcoord_cache[c] = [pcs[int(i)][int(s)] for i, s in zip(its, segs)]
        self.cluster_structures = cluster_structures
        self.cluster_structure_weights = cluster_structure_weights
        self.pcoord_cache = pcoord_cache
        self.structure_iteration_segments = structure_iteration_segments

    # ------------------------------------------------------------------------------------------
    def organize_stratified(self, use_ray=True, progress_bar=None):
        """`_clustering.py:920-1142`: keep the largest strongly-connected set, delete the other
        centroids, re-discretize, re-build the flux matrix, sort states by mean pcoord."""
        # the raw flux matrix stays where get_fluxMatrix left it (HBM): only its sparsity pattern travels, as CSR
        h = self.h
        raw_dev, raw_host = self._matrix_device("fluxMatrixRaw"), self.__dict__.get("_h_fluxMatrixRaw")
        ns0 = int(raw_dev.shape[0])
        indptr, indices = h.pattern_csr(raw_dev, ns0)
        connected_sets = connected_sets_from_pattern(indptr, indices, ns0, extra_edge=(ns0 - 1, ns0 - 2))  # recycling edge (:938)
        start_cleaning_idx = 1
        if len(connected_sets) == 1:
            states_to_remove = np.array([], dtype=int)
        else:
            states_to_remove = np.concatenate(connected_sets[start_cleaning_idx:])

        cl = self.clusters
        pre = cl.n_fitted()  # :972-977
        S = cl.bin_mapper.nbins
        empty_we_bins = set()
        for we_bin in range(S):
            offset = sum(pre[:we_bin])
            in_bin = range(offset, offset + pre[we_bin])
            to_clean = np.intersect1d(states_to_remove, in_bin)
            if len(to_clean) == 0:
                if len(in_bin) == 0:
                    empty_we_bins.add(we_bin)
                continue
            elif len(to_clean) == len(in_bin):
                empty_we_bins.add(we_bin)
            m = cl.cluster_models[we_bin]
            m.cluster_centers_ = np.delete(m.cluster_centers_, to_clean - offset, 0)  # :1041-1045

        log.info(f"Started with {self.n_clusters} clusters, and removed {len(states_to_remove)}")
        self.n_clusters = self.n_clusters - len(states_to_remove)  # :1050 (quirk Q9)
        assert self.n_clusters > 1, "All clusters would be cleaned! You probably need more data, fewer clusters, or both."

        populated = np.setdiff1d(range(S), list(empty_we_bins))
        for eb in empty_we_bins:  # :1070-1078
            cl.we_remap[eb] = int(self.find_nearest_bin(cl.bin_mapper, eb, populated))
        for we_bin in range(S):  # :1081-1094 every bin must now resolve to fitted centroids
            tgt = cl.cluster_models[cl.we_remap[we_bin]]
            if not tgt.fitted:
                raise AttributeError(f"Error obtaining clusters for WE bin {we_bin}, remapped to {cl.we_remap[we_bin]}")

        cl.toggle = False
        cl.processing_from = False
        self.launch_ray_discretization(progress_bar=progress_bar)  # :1101-1104
        pcoord_sort_indices = self.get_cluster_centers()            # :1106
        self.get_fluxMatrix(*self._fluxMatrixParams, use_ray=use_ray, progress_bar=progress_bar)  # :1108-1115

        # `fluxMatrix = fluxMatrixRaw[perm][:, perm] / np.sum(...)` (:1117-1121) on the device: gather, numpy's own pairwise
        # sum (hamsm_np_sum reproduces np.sum bit for bit), true division
        n = self.n_clusters + 2
        d_new = self._matrix_device("fluxMatrixRaw")
        d_org = h.alloc((n, n), np.float64)
        # numpy lays `a[perm, :][:, perm]` out column-major and np.sum walks memory: the sum is taken over the transpose
        h.permute_scale(d_new, n, pcoord_sort_indices, d_org, 1.0, transpose=True)
        total = h.np_sum(d_org, n * n)
        h.permute_scale(d_new, n, pcoord_sort_indices, d_org, total)
        self._set_matrix_device("fluxMatrix", d_org)
        self.__dict__["_d_fluxMatrixRaw"], self.__dict__["_h_fluxMatrixRaw"] = raw_dev, raw_host  # :1123 the pre-clean matrix
        self.indBasis = np.array([self.n_clusters])
        self.indTargets = np.array([self.n_clusters + 1])
        self.nBins = self.n_clusters + 2
        self.update_sorted_cluster_centers()
        self.cluster_mapping = {x: x for x in range(self.n_clusters + 2)}

        indptr, indices = h.pattern_csr(d_org, n)
        assert len(connected_sets_from_pattern(indptr, indices, n, extra_edge=(n - 1, n - 2))[start_cleaning_idx:]) == 0, \
            "Still not clean after cleaning!"

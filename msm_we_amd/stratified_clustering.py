"""Drop-in for ``msm_we.stratified_clustering.StratifiedClusters`` (reference
``msm_we/stratified_clustering.py:6-212``) whose arithmetic runs in the HIP library.

Level B3 of the boundary: ``.predict(X)``, ``.cluster_models[i].cluster_centers_``
(get/set), ``.cluster_models[i].partial_fit``, ``.we_remap``, ``.bin_mapper``, ``.target_bins`` /
``.basis_bins``, ``.processing_from`` / ``.toggle`` behave as in the reference.  The per-stratum
model mirrors ``sklearn.cluster.MiniBatchKMeans.partial_fit`` (sklearn 1.7.2
``cluster/_kmeans.py:2203-2301``): every random draw is made here, on the host, with NumPy's legacy
``RandomState`` in sklearn's call order, and handed to the library (include/hamsm.h).

State ownership: a ctx holds ONE set of centroids.  Several ``StratifiedClusters`` objects (the
model, its ``post_cluster_model`` copy, block-validation copies) may share a ctx; each keeps its
centroids as plain NumPy on the host and ``_activate()`` pushes them when it becomes the ctx's
active owner.  While training, the device copy is authoritative and is pulled back lazily.
Objects pickle / deepcopy as plain NumPy (no live handles), as the restart plugin requires.
"""
from __future__ import annotations

import numbers
import weakref

import numpy as np

from . import _abi

__all__ = ["StratifiedClusters", "StratumKMeans"]


class _NoEntropy(np.random.bit_generator.ISeedSequence):
    """seed sequence that hashes nothing: the MT19937 built on it is re-seeded at once, the legacy way"""

    def generate_state(self, n_words, dtype=np.uint32):
        return np.zeros(n_words, dtype=dtype)


_NO_ENTROPY = _NoEntropy()


def _random_state_from_int(seed):
    """`np.random.RandomState(seed)`, state for state (test_oracle_golden.py), at under half its price: RandomState(int) first
    builds its MT19937 from a SeedSequence (entropy pool, hashing: most of the 60-130 us) and then throws that state away in
    `_legacy_seeding(seed)`.  Four of these per fit stand between the prepass and the seeding launch, with the device idle."""
    rs = np.random.RandomState(np.random.MT19937(_NO_ENTROPY))
    rs.seed(seed)
    return rs


def _check_random_state(seed, owner=None):
    # sklearn.utils.check_random_state semantics (utils/validation.py).  (Round 3 recycled the RandomState objects of models that
    # had been replaced, deciding ownership by reference counts -- 0.1 ms per construction saved, fragile under any caller that
    # held one: dropped in round 4.)
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, numbers.Integral):
        return _random_state_from_int(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError(f"{seed!r} cannot be used to seed a numpy.random.RandomState instance")


def _csq_like_sklearn(Cm):
    # row_norms(centers, squared=True) == einsum("ij,ij->i") (sklearn utils/extmath.py; used at
    # _k_means_lloyd.pyx:96): same call => same bits as the reference's predict on this host
    return np.einsum("ij,ij->i", Cm, Cm)


class StratumKMeans:
    """MiniBatchKMeans of one WE bin."""

    def __init__(self, owner, stratum, n_clusters=8, *, random_state=None, batch_size=1024, init_size=None,
                 reassignment_ratio=0.01, max_iter=100, **_ignored):
        d = self.__dict__
        d["_owner"] = owner
        d["stratum"] = int(stratum)
        d["n_clusters"] = int(n_clusters)
        d["random_state"] = random_state
        d["batch_size"] = batch_size
        d["init_size"] = init_size
        d["reassignment_ratio"] = reassignment_ratio
        d["max_iter"] = max_iter
        d["_host_centers"] = None   # (K_s, D) float64, or None = never fitted
        d["_host_counts"] = None
        d["_device_newer"] = False  # device copy is ahead of _host_centers (after a training step)
        d["_host_newer"] = False    # host copy must be pushed before the next device op
        d["_counts_cache"] = None

    @property
    def _h(self) -> "_abi.Hamsm":
        return self._owner._h

    @property
    def fitted(self) -> bool:
        return self._host_centers is not None or self._device_newer

    def _pull(self):
        if self._device_newer:
            Cm, counts, _ = self._h.get_centroids(self.stratum)
            self.__dict__["_host_centers"] = Cm
            self.__dict__["_host_counts"] = counts
            self.__dict__["_device_newer"] = False

    def _push(self):
        h = self._h
        Cm = self._host_centers
        if Cm is None or len(Cm) == 0:
            h.set_centroids(self.stratum, None)
        else:
            h.set_centroids(self.stratum, Cm, csq=_csq_like_sklearn(Cm), counts=self._host_counts
                            if self._host_counts is not None and len(self._host_counts) == len(Cm) else None)
        self.__dict__["_host_newer"] = False

    def __getattr__(self, name):
        # `hasattr(model, "cluster_centers_")` is how the reference asks "was this bin ever fitted"
        # (stratified_clustering.py:146,181,187; _clustering.py:974,997)
        if name == "cluster_centers_":
            if not self.fitted:
                raise AttributeError(name)
            self._pull()
            return self._host_centers
        if name == "_counts":
            if not self.fitted:
                raise AttributeError(name)
            self._pull()
            return self._host_counts
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name == "cluster_centers_":
            # cleaning deletes rows (_clustering.py:1041-1045)
            self._pull()
            self.__dict__["_host_centers"] = np.ascontiguousarray(value, dtype=np.float64)
            self.__dict__["_host_counts"] = None
            self.__dict__["_host_newer"] = True
            self.__dict__["_counts_cache"] = None
            return
        self.__dict__[name] = value

    def __getstate__(self):
        if self._device_newer and self._owner._h is not None:
            self._pull()
        d = dict(self.__dict__)
        d["_owner"] = None
        d["_device_newer"] = False
        d["_host_newer"] = True
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)

    # -- sklearn MiniBatchKMeans.partial_fit on a device-resident batch --------------------------
    def partial_fit_device(self, d_X, x_dtype, ldx, d_rows, n_b, d_w=None, w_rows_host=None):
        """Batch = rows ``d_rows[0:n_b]`` of ``d_X`` (device).  ``w_rows_host``: the batch's sample
        weights on the host when weights are used (needed for the k-means++ ``choice(p=...)`` draw)."""
        self._owner._activate()
        h = self._h
        K = self.n_clusters
        if "_random_state" not in self.__dict__:
            self._random_state = _check_random_state(self.random_state, self)  # _kmeans.py:2237-2239
        rs = self._random_state
        if not self.fitted:
            if n_b < K:
                raise ValueError(f"n_samples={n_b} should be >= n_clusters={K}.")  # _kmeans.py:875-879
            # _check_params_vs_input, _kmeans.py:1922-1943
            self._batch_size = min(self.batch_size, n_b)
            init_size = self.init_size
            if init_size is None:
                init_size = 3 * self._batch_size
                if init_size < K:
                    init_size = 3 * K
            elif init_size < K:
                init_size = 3 * K
            self._init_size = min(init_size, n_b)
            # _init_centroids, _kmeans.py:1003-1011
            sub = None
            n = n_b
            w_host = np.ones(n_b) if w_rows_host is None else np.asarray(w_rows_host, dtype=np.float64)
            if self._init_size < n_b:
                sub = rs.randint(0, n_b, self._init_size)
                n = self._init_size
                w_host = w_host[sub]
            # _kmeans_plusplus, _kmeans.py:225,243
            first = rs.choice(n, p=w_host / w_host.sum())
            L = 2 + int(np.log(K))
            u = rs.uniform(size=(K - 1) * L)  # K-1 calls of uniform(size=L): the same stream as one call
            h.kmeans_seed(self.stratum, d_X, x_dtype, ldx, d_rows, n_b, d_w, sub, K, first, u, L)
            self._counts_cache = np.zeros(K)  # _kmeans.py:2271
            self._n_since_last_reassign = 0   # :2274
            self.__dict__["_device_newer"] = True
        return self._host_step(d_X, x_dtype, ldx, d_rows, n_b, d_w)

    def _host_step(self, d_X, x_dtype, ldx, d_rows, n_b, d_w=None):
        """One `_mini_batch_step` with the host in the loop: the reassignment decision and its RandomState draw
        (`_kmeans.py:1643-1673`) need the counts on the host."""
        h = self._h
        K = self.n_clusters
        rs = self._random_state
        # _random_reassign, _kmeans.py:2029-2043 (evaluated before the step)
        self._n_since_last_reassign += self._batch_size
        counts = self._counts_cache if self._counts_cache is not None else h.get_counts(self.stratum)
        do_reassign = bool((counts == 0).any() or self._n_since_last_reassign >= 10 * K)
        if do_reassign:
            self._n_since_last_reassign = 0
        h.minibatch_step(self.stratum, d_X, x_dtype, ldx, d_rows, n_b, d_w)
        self.__dict__["_device_newer"] = True
        self._counts_cache = None  # fetched lazily (one small D2H per step)
        self._last_n_re = 0
        if do_reassign and self.reassignment_ratio > 0:
            self._host_reassign(d_X, x_dtype, ldx, d_rows, n_b)
        self.n_steps_ = self.__dict__.get("n_steps_", 0) + 1
        return self

    def _host_reassign(self, d_X, x_dtype, ldx, d_rows, n_b):
        """`_mini_batch_step`, `_kmeans.py:1643-1673`, for the step just applied on the device."""
        h = self._h
        weight_sums = h.get_counts(self.stratum)
        to_reassign = weight_sums < self.reassignment_ratio * weight_sums.max()
        if to_reassign.sum() > 0.5 * n_b:
            keep = np.argsort(weight_sums)[int(0.5 * n_b):]
            to_reassign[keep] = False
        n_re = int(to_reassign.sum())
        if n_re:
            new_rows = self._random_state.choice(n_b, replace=False, size=n_re)
            new_count = float(np.min(weight_sums[~to_reassign]))
            h.reassign(self.stratum, d_X, x_dtype, ldx, d_rows, np.where(to_reassign)[0], new_rows, new_count)
            weight_sums[to_reassign] = new_count
        self._counts_cache = weight_sums
        self._last_n_re = n_re
        return n_re

    # -- host-array entry point used by the reference's training loop (_clustering.py:909-911) ---
    def partial_fit(self, X, y=None, sample_weight=None):
        h = self._h
        X = np.ascontiguousarray(X)
        if X.dtype != np.float32:
            X = np.ascontiguousarray(X, dtype=np.float64)
        dt = _abi.F32 if X.dtype == np.float32 else _abi.F64
        if X.ndim != 2 or X.shape[1] != h.D:
            raise ValueError(f"X has shape {X.shape}, expected (n, {h.D})")
        d_X = h.to_device(X)
        d_w = None
        w_host = None
        if sample_weight is not None:
            w_host = np.ascontiguousarray(sample_weight, dtype=np.float64)
            d_w = h.to_device(w_host)
        try:
            self.partial_fit_device(d_X, dt, h.D, None, X.shape[0], d_w, w_host)
        finally:
            d_X.free()
            if d_w is not None:
                d_w.free()
        return self


class _RemapDict(dict):
    """we_remap is a plain dict in the reference, written item-wise (_clustering.py:730,1078)."""

    def __init__(self, owner, *a):
        super().__init__(*a)
        self._owner = owner

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        self._owner.__dict__["_remap_dirty"] = True

    def __reduce__(self):
        return (dict, (dict(self),))


class StratifiedClusters:
    """See module docstring.  ``model`` supplies ``pcoord0List/pcoord1List`` like the reference's
    modelWE (stratified_clustering.py:129-132); basis/target tests use the bounds pushed to the ctx."""

    def __init__(self, bin_mapper, model, n_clusters, target_bins=(), *, h: "_abi.Hamsm" = None, **_cluster_args):
        cluster_args = {"n_clusters": n_clusters, "max_iter": 100}  # stratified_clustering.py:55-59
        cluster_args.update(_cluster_args)
        self.n_clusters_per_bin = n_clusters
        self.bin_mapper = bin_mapper
        self.n_total_clusters = n_clusters * (bin_mapper.nbins - len(target_bins))
        self.cluster_args = cluster_args
        self.model = model
        self._h = h
        self.cluster_models = [StratumKMeans(self, s, **cluster_args) for s in range(bin_mapper.nbins)]
        self.processing_from = False
        self.toggle = False
        self._remap = _RemapDict(self, {x: x for x in range(bin_mapper.nbins)})
        self._remap_dirty = True
        self.legitimate_bins = range(bin_mapper.nbins)
        self.target_bins = set()
        self.basis_bins = set()

    @property
    def we_remap(self):
        return self._remap

    @we_remap.setter
    def we_remap(self, d):
        self._remap = _RemapDict(self, d)
        self._remap_dirty = True

    # -- ctx ownership ----------------------------------------------------------------------
    def _pull_all(self):
        for m in self.cluster_models:
            m._pull()

    def _activate(self):
        """Make this object's centroids / remap the ones in the ctx."""
        h = self._h
        if h is None:
            raise RuntimeError("StratifiedClusters has no device context attached (use attach(h))")
        prev = getattr(h, "_active_clusters", None)
        prev = prev() if prev is not None else None
        if prev is not self:
            if prev is not None:
                prev._pull_all()
            for m in self.cluster_models:
                if not m._device_newer or prev is not None:
                    m._push()
            h._active_clusters = weakref.ref(self)
            self._remap_dirty = True
        else:
            for m in self.cluster_models:
                if m._host_newer:
                    m._push()
        if self._remap_dirty:
            h.set_remap([self._remap[b] for b in range(self.bin_mapper.nbins)])
            self._remap_dirty = False

    def _adopt_device_state(self):
        """After a model broadcast the ctx holds the authoritative centroids: become their owner."""
        h = self._h
        for m in self.cluster_models:
            k = h.num_centroids(m.stratum)
            m.__dict__["_host_centers"] = None
            m.__dict__["_host_counts"] = None
            m.__dict__["_device_newer"] = k > 0
            m.__dict__["_host_newer"] = False
        h._active_clusters = weakref.ref(self)
        self._remap_dirty = True

    def attach(self, h: "_abi.Hamsm"):
        """Bind an unpickled / copied object to a context."""
        self._h = h
        for m in self.cluster_models:
            m.__dict__["_owner"] = self
        return self

    def n_fitted(self):
        """len(cluster_centers_) per stratum, 0 where unfitted (stratified_clustering.py:143-150)."""
        return [len(m.cluster_centers_) if m.fitted else 0 for m in self.cluster_models]

    def predict(self, coords):
        iter_pcoords = self.model.pcoord0List if self.processing_from else self.model.pcoord1List
        self._activate()
        if getattr(self.model, "_mapper_external", False):  # the mapper's own assign (stratified_clustering.py:134)
            self._h.set_frame_bins_host(np.asarray(self.bin_mapper.assign(np.asarray(iter_pcoords)), dtype=np.int64))
        labels, seen = self._h.predict_host(np.asarray(coords), np.asarray(iter_pcoords), want_seen=True)
        self.target_bins.update(np.nonzero(seen[0])[0].tolist())
        self.basis_bins.update(np.nonzero(seen[1])[0].tolist())
        if self.toggle:  # stratified_clustering.py:206-210
            self.processing_from = not self.processing_from
        return labels.astype(np.int64)

    # -- pickling / deepcopy: plain NumPy state, no live handles ---------------------------------
    def __getstate__(self):
        if self._h is not None:
            self._pull_all()
        d = dict(self.__dict__)
        d["_h"] = None
        d["_remap"] = dict(self._remap)
        d["model"] = None if d.get("model") is not None and getattr(d["model"], "_pickle_skip", False) else d["model"]
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._remap = _RemapDict(self, self._remap)
        self._remap_dirty = True
        for m in self.cluster_models:
            m.__dict__["_owner"] = self

    def __deepcopy__(self, memo):
        import copy

        if self._h is not None:
            self._pull_all()
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_h":
                new.__dict__[k] = v  # copies share the ctx, not the centroids
            elif k == "_remap":
                new.__dict__[k] = _RemapDict(new, dict(v))
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        for m in new.cluster_models:
            m.__dict__["_owner"] = new
            m.__dict__["_host_newer"] = True
            m.__dict__["_device_newer"] = False
        new._remap_dirty = True
        return new

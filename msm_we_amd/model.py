"""``modelWE``: host-side mirror of the reference's model class for the haMSM construction path.

Method names, arguments, attributes and error behaviour follow ``msm_we.modelWE`` (reference
``msm_we/msm_we.py``, mixins in ``msm_we/_hamsm/``) so that a user of the reference finds the same
calls; the arithmetic of every step on the hot path is executed by the HIP library
through :mod:`msm_we_amd._abi`.  What stays on the host is what the reference's north star keeps
there: control flow, RandomState draws, strongly-connected components and the sparse solve.

Data enters through the same five loader hooks the reference reads HDF5 with (``load_iter_data``,
``load_iter_coordinates``, ``load_iter_coordinates0``, ``get_transition_data_lag0``,
``get_iterations``; ``_data.py``) -- see :mod:`msm_we_amd.data` for the in-memory and WESTPA-HDF5
sources.  All iterations are staged once into HBM (``stage_iterations``) and stay resident across
train -> discretize -> flux -> clean -> re-discretize -> re-flux and block validation.
"""
from __future__ import annotations

import logging
from copy import deepcopy

import numpy as np

from . import _abi
from .analysis import AnalysisMixin
from .clustering import ClusteringMixin
from .data import DataMixin
from .fluxmatrix import FluxMatrixMixin

log = logging.getLogger("msm_we_amd")

__all__ = ["modelWE"]


class modelWE(DataMixin, ClusteringMixin, FluxMatrixMixin, AnalysisMixin):
    """Reference: ``msm_we/msm_we.py:35-42`` (class assembled from mixins)."""

    class BlockValidationError(Exception):
        pass

    def __init__(self, h: "_abi.Hamsm" = None, device: int = 0):
        # msm_we.py:63-141
        self.modelName = None
        self._n_lag = 0
        self.pcoord_ndim = None
        self.pcoord_len = None
        self.tau = None
        self.target_bin_center = None  # singular: never set by the reference either (quirk Q8)
        self.basis_bin_center = None
        self._basis_pcoord_bounds = None
        self._target_pcoord_bounds = None
        self.nAtoms = None
        self.coordinates = None
        self.ndim = None
        self.targetRMSD_centers = None
        self.indBasis = None
        self.clustering_method = None
        self.validation_models = []
        self.pre_discretization_model = None
        self.use_weights_in_clustering = False
        self.dimReduceMethod = "none"
        self._h = h
        self._device = device
        self._store = None
        self._mapper_external = False
        self.seg_weights = {}

    # -- device context ---------------------------------------------------------------------
    @property
    def h(self) -> "_abi.Hamsm":
        if self._h is None:
            self._h = _abi.Hamsm(device=self._device)  # raises if the HIP library is not built
        return self._h

    def __deepcopy__(self, memo):
        # validation models share the device context and the staged iterations (read-only), but own
        # their clusters / labels (reference: deepcopy(self) at _clustering.py:195, msm_we.py:941)
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_h", "_store", "_train_store", "_source", "_d_parent", "_d_child"):
                new.__dict__[k] = v  # shared, read-only: discretization allocates fresh label buffers
            elif k in ("_dtrajs_cache", "_pairs_cache"):
                new.__dict__[k] = None  # host views of the shared device labels: rebuilt on first access
            else:
                new.__dict__[k] = deepcopy(v, memo)
        return new

    # -- pickling (the reference pickles the model after every stage: tests/fixtures/hamsms.py:13-56, the restart
    #    plugin, ray.put at _clustering.py:1165-1176).  The pickle holds plain NumPy / Python state: no context, no
    #    device buffers; the discretization is materialised (`dtrajs`, `pair_dtrajs` as host arrays).  After loading,
    #    `attach(h)` binds a context; frames are re-staged from the iteration source on first use. -----------------------
    _DEVICE_STATE = ("_h", "_store", "_train_store", "_d_parent", "_d_child", "_dtrajs_cache", "_pairs_cache")

    def __getstate__(self):
        for name in ("fluxMatrixRaw", "fluxMatrix"):   # device-resident matrices: copied out before the handles are dropped
            if self.__dict__.get("_d_" + name) is not None:
                getattr(self, name)
        d = dict(self.__dict__)
        d["_d_fluxMatrixRaw"] = d["_d_fluxMatrix"] = None
        if d.get("_d_child") is not None:
            d["_dtrajs_plain"], d["_pairs_plain"] = self._labels_host(0), self._labels_host(1)
        for k in self._DEVICE_STATE:
            if k in d:
                d[k] = None
        if d.get("_featurizer_spec") is not None:
            d["_featurizer"] = None   # travels by name (`set_featurizer`), resolved again on first use
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        cl = self.__dict__.get("clusters")
        if cl is not None and getattr(cl, "model", None) is None:
            cl.model = self

    def attach(self, h: "_abi.Hamsm" = None):
        """Bind an unpickled model (and its cluster object, validation copies) to a device context; the default creates
        one on `self._device`.  Geometry, mapper and bounds are pushed again, centroids follow on the first device op."""
        if h is None:
            h = _abi.Hamsm(device=self._device)
        self._h = h
        seen = {id(self)}
        for m in [getattr(self, "post_cluster_model", None), getattr(self, "pre_discretization_model", None),
                  *(getattr(self, "validation_models", None) or [])]:
            if m is not None and id(m) not in seen and hasattr(m, "attach"):
                seen.add(id(m))
                m.attach(h)
        cl = getattr(self, "clusters", None)
        if cl is not None:
            cl.attach(h)
            if getattr(self, "_source", None) is not None and getattr(self, "maxIter", None):
                store = self.stage_iterations()
                self._configure_device(cl.bin_mapper, store.D)
        return self

    # -- msm_we.py:143-277 ------------------------------------------------------------------
    def initialize(self, fileSpecifier, refPDBfile, modelName, basis_pcoord_bounds=None, target_pcoord_bounds=None,
                   dim_reduce_method="none", tau=None, pcoord_ndim=1, auxpath="coord",
                   _suppress_boundary_warning=False, use_weights_in_clustering=False):
        self.modelName = modelName
        self.fileList = fileSpecifier if isinstance(fileSpecifier, list) else str(fileSpecifier).split(" ")
        self.n_data_files = len(self.fileList)
        self.pcoord_ndim = pcoord_ndim
        self.pcoord_len = 2
        if basis_pcoord_bounds is not None:
            self.basis_pcoord_bounds = basis_pcoord_bounds
        if target_pcoord_bounds is not None:
            self.target_pcoord_bounds = target_pcoord_bounds
        self.auxpath = auxpath
        if tau is None:
            log.warning("No tau provided, defaulting to 1.")
            tau = 1.0
        self.tau = float(tau)
        self.set_topology(refPDBfile)
        self.dimReduceMethod = "pca" if dim_reduce_method is None else dim_reduce_method
        try:
            self.load_iter_data(1)
            self.load_iter_coordinates0()
            self.coordsExist = True
        except KeyError:
            self.coordsExist = False
        self.use_weights_in_clustering = use_weights_in_clustering

    def set_topology(self, topology):
        """msm_we.py:1011-1075.  The dict form carries what the hot path needs (`nAtoms`, `coord_ndim`, `coords`); a file
        path (PDB / .dat, parsed with mdtraj in the reference -- "should no longer be necessary", hamsm_driver.py:90) or
        None leaves the atom count to be read off the first iteration's coordinates."""
        if isinstance(topology, dict):
            self.reference_coord = topology["coords"]
            self.nAtoms = topology["nAtoms"]
            self.coord_ndim = topology["coord_ndim"]
            return
        src = getattr(self, "_source", None)
        if src is None:
            raise NotImplementedError("pass the topology as dict(coords=, nAtoms=, coord_ndim=), or set an iteration source first")
        d = src.load(src.iterations()[0])
        c = np.asarray(d["child"] if "child" in d else np.asarray(d["Xc"])[:, :, None])
        self.reference_structure_file = topology
        self.reference_coord = None
        self.nAtoms = int(c.shape[1])
        self.coord_ndim = int(c.shape[2]) if c.ndim > 2 else 1

    # msm_we.py:300-347 / 397-444
    @staticmethod
    def _check_bounds(bounds, ndim):
        bounds = np.array(bounds, dtype=float)
        if bounds.ndim == 1:
            bounds = bounds.reshape(1, 2)
        assert bounds.shape == (ndim, 2), f"Shape of bounds was {bounds.shape}, should've been ({ndim}, 2)"
        assert np.all([b[0] < b[1] for b in bounds]), "A boundary has a lower bound larger than its upper bound"
        centers = np.full(ndim, np.nan)
        for i, (lo, hi) in enumerate(bounds):
            if not abs(lo) == np.inf and not abs(hi) == np.inf:
                centers[i] = np.mean([lo, hi])
            else:
                centers[i] = [lo, hi][int(abs(lo) == np.inf)]
        return bounds, centers

    @property
    def basis_pcoord_bounds(self):
        return self._basis_pcoord_bounds

    @basis_pcoord_bounds.setter
    def basis_pcoord_bounds(self, bounds):
        self._basis_pcoord_bounds, self.basis_bin_centers = self._check_bounds(bounds, self.pcoord_ndim)

    @property
    def target_pcoord_bounds(self):
        return self._target_pcoord_bounds

    @target_pcoord_bounds.setter
    def target_pcoord_bounds(self, bounds):
        self._target_pcoord_bounds, self.target_bin_centers = self._check_bounds(bounds, self.pcoord_ndim)

    @property
    def n_lag(self):
        return self._n_lag

    @n_lag.setter
    def n_lag(self, lag):
        if not lag == 0:  # msm_we.py:353-362
            raise NotImplementedError("Only a lag of 1 tau (n_lag = 0) is currently supported")
        self._n_lag = lag

    # msm_we.py:462-527 -- host copies for callers outside the hot path; the library applies the
    # same strict f64 tests per frame inside assign / flux (hamsm_classify bit0/bit1)
    def is_WE_basis(self, pcoords):
        pcoords = np.asarray(pcoords)
        b = self.basis_pcoord_bounds
        return np.all((pcoords[:, : self.pcoord_ndim] > b[:, 0]) & (pcoords[:, : self.pcoord_ndim] < b[:, 1]), axis=1)

    def is_WE_target(self, pcoords):
        pcoords = np.asarray(pcoords)
        b = self.target_pcoord_bounds
        return np.all((pcoords[:, : self.pcoord_ndim] > b[:, 0]) & (pcoords[:, : self.pcoord_ndim] < b[:, 1]), axis=1)

    # _dimensionality.py:110-345 -- only "none" (identity) is on the path this round
    class Coordinates:
        def transform(self, coords):
            return coords

    def dimReduce(self, first_iter=1, first_rough_iter=None, last_iter=None, rough_stride=10, fine_stride=1,
                  variance_cutoff=0.95, use_weights=True, progress_bar=None):
        """`_dimensionality.py:110-241`.  "none": identity.  "pca": the reference's streaming scheme with the same
        estimator (scikit-learn's IncrementalPCA) -- a rough pass over every `rough_stride`-th iteration to find the
        number of components that explain `variance_cutoff`, then the fit with that many components over all
        iterations -- on the host; the fitted map is applied to every frame on the device at staging time
        (`hamsm_project`).  TICA / VAMP need deeptime (not in the image)."""
        with self._source_session():    # one pass (two for PCA) over the source: its files are open for that long only
            return self._dimReduce(first_iter, first_rough_iter, last_iter, rough_stride, fine_stride, variance_cutoff)

    def _dimReduce(self, first_iter, first_rough_iter, last_iter, rough_stride, fine_stride, variance_cutoff):
        if self.dimReduceMethod == "none":
            self.ndim = int(3 * self.nAtoms)
            self.coordinates = self.Coordinates()
            return
        if self.dimReduceMethod != "pca":
            raise NotImplementedError(f"dimReduceMethod {self.dimReduceMethod!r} is not built (needs deeptime; SURVEY §8f-3)")
        from sklearn.decomposition import IncrementalPCA as iPCA

        if last_iter is None:
            last_iter = self.maxIter
        if first_rough_iter is None:
            total = last_iter
            rough_iters = range(1, total, rough_stride) if total > 100 else range(total // 2, total)  # :150-158
        else:
            rough_iters = range(first_rough_iter, last_iter, rough_stride)
        if getattr(self, "pca_fit", "host") == "device":
            return self._dimReduce_device(rough_iters, range(first_iter, last_iter, fine_stride), variance_cutoff,
                                          (last_iter - first_iter) // fine_stride - 1)
        rough_ipca = iPCA()
        for iteration in rough_iters:  # do_pca, :69-82
            iter_coords = self.get_iter_coordinates(iteration)
            if iter_coords.shape[0] == 0:
                continue
            rough_ipca.partial_fit(self.processCoordinates(iter_coords))
        components_for_var = int(np.argmax(np.cumsum(rough_ipca.explained_variance_ratio_) > variance_cutoff) + 1)  # :192-197
        components_for_var = min(components_for_var, (last_iter - first_iter) // fine_stride - 1)
        ipca = iPCA(n_components=components_for_var)
        extra = 0
        for iteration in range(first_iter, last_iter, fine_stride):  # do_full_pca, :84-108 / :206-235
            if extra > 0:
                extra -= 1
                continue
            iter_coords = self.get_iter_coordinates(iteration)
            used = 0
            while iter_coords.shape[0] <= components_for_var:  # keep adding iterations until there are enough frames
                used += 1
                more = self.get_iter_coordinates(iteration + used)
                if more.shape[0] == 0:
                    continue
                iter_coords = np.append(iter_coords, more, axis=0)
            ipca.partial_fit(self.processCoordinates(iter_coords))
            extra = used
        self.coordinates = ipca
        self.ndim = components_for_var

    pca_fit = "host"  # "device": exact PCA from covariance sums accumulated in HBM (pca.py); "host": the reference's IncrementalPCA

    def _pca_sums(self, iterations):
        """the features of `iterations` (what `do_pca` hands to partial_fit, _dimensionality.py:69-82) streamed through
        hamsm_pca_accumulate: -> (n, shift, S1, S2)"""
        h = self.h
        buf, cap, started = None, 0, False
        try:
            for it in iterations:
                iter_coords = self.get_iter_coordinates(it)
                if iter_coords.shape[0] == 0:
                    continue
                X = np.ascontiguousarray(self.processCoordinates(iter_coords), dtype=np.float64)
                n, D = X.shape
                if not started:
                    h.pca_begin(D)
                    started = True
                if n * D > cap:
                    if buf is not None:
                        buf.free()
                    cap = max(n * D, 2 * cap)
                    buf = h.alloc(cap, np.float64)
                h.sync()                      # the previous block has been consumed
                h.h2d(buf, X)
                h.pca_accumulate(buf, _abi.F64, D, n)
            if not started:
                raise ValueError("no coordinates to fit the PCA on")
            return h.pca_finish()
        finally:
            if buf is not None:
                buf.free()

    def _dimReduce_device(self, rough_iters, fine_iters, variance_cutoff, comp_cap):
        """the reference's two-stage scheme (`_dimensionality.py:139-241`) with exact PCA in both stages: the rough subset decides
        how many components explain `variance_cutoff` (:192-200), all iterations give the components"""
        from .pca import CovariancePCA

        rough = CovariancePCA().fit_from_sums(*self._pca_sums(rough_iters))
        components_for_var = int(np.argmax(np.cumsum(rough.explained_variance_ratio_) > variance_cutoff) + 1)
        components_for_var = min(components_for_var, comp_cap)
        self.coordinates = CovariancePCA(n_components=components_for_var).fit_from_sums(*self._pca_sums(fine_iters))
        self.ndim = components_for_var

    # msm_we.py:588-882, without the rich table / Ray bootstrap
    def build_analyze_model(self, file_paths, ref_struct, modelName, basis_pcoord_bounds, target_pcoord_bounds,
                            dimreduce_method, tau, n_clusters, stratified=True, streaming=True, use_ray=True,
                            fluxmatrix_iters=(1, -1), fluxmatrix_iters_to_use=None, cross_validation_groups=2,
                            cross_validation_blocks=4, show_live_display=False, allow_validation_failure=False,
                            step_kwargs=None, max_coord_iter=-1, **_ignored):
        step_kwargs = step_kwargs or {}
        self.initialize(file_paths, ref_struct, modelName, basis_pcoord_bounds, target_pcoord_bounds,
                        dimreduce_method, tau, **step_kwargs.get("initialize", {}))
        self.get_iterations()
        self.get_coordSet(self.maxIter if max_coord_iter == -1 else max_coord_iter)
        self.dimReduce(**step_kwargs.get("dimReduce", {}))
        self.cluster_coordinates(n_clusters=n_clusters, streaming=streaming, use_ray=use_ray, stratified=stratified,
                                 store_validation_model=cross_validation_groups > 0,
                                 **step_kwargs.get("clustering", {}))
        fi = list(fluxmatrix_iters)
        if fi[1] == -1:
            fi[1] = self.maxIter
        self.get_fluxMatrix(n_lag=0, first_iter=fi[0], last_iter=fi[1], iters_to_use=fluxmatrix_iters_to_use,
                            use_ray=use_ray, **step_kwargs.get("fluxmatrix", {}))
        self.organize_fluxMatrix(use_ray=use_ray, **step_kwargs.get("organize", {}))
        self.get_Tmatrix()
        self.get_steady_state()
        self.get_steady_state_target_flux()
        if cross_validation_groups > 0:
            try:
                self.do_block_validation(cross_validation_groups=cross_validation_groups,
                                         cross_validation_blocks=cross_validation_blocks, use_ray=use_ray,
                                         **step_kwargs.get("block_validation", {}))
            except Exception as e:  # noqa: BLE001  (msm_we.py:865-872)
                log.error(e)
                if not allow_validation_failure:
                    raise

    # msm_we.py:884-1009
    def do_block_validation(self, cross_validation_groups, cross_validation_blocks, use_ray=True, progress_bar=None):
        """Split iterations into uniform blocks, deal blocks round-robin into groups, and rebuild
        flux matrix -> clean -> T -> steady state -> target flux per group on a copy of the
        post-clustering model.  Copies share the ctx and the staged frames; nothing is re-uploaded."""
        assert getattr(self, "post_cluster_model", None) is not None, (
            "Perform clustering with cluster_coordinates() before attempting"
            "block validation -- self.post_cluster_model is not set.")
        validation_models = [deepcopy(self.post_cluster_model) for _ in range(cross_validation_groups)]
        maxIter = self.post_cluster_model.maxIter
        per_block = maxIter // cross_validation_blocks  # :924
        block_iterations = [[a, a + per_block] for a in range(1, maxIter, per_block)]
        block_iterations[-1][-1] = block_iterations[-1][-1] - 1  # :931
        group_blocks = [range(g, cross_validation_blocks, cross_validation_groups) for g in range(cross_validation_groups)]
        validation_iterations = []
        for group in range(cross_validation_groups):
            its = []
            for block in group_blocks[group]:
                its.extend(range(*block_iterations[block]))
            validation_iterations.append(its)
            try:
                m = validation_models[group]
                m.clusters.model = m
                m.get_fluxMatrix(0, iters_to_use=its, use_ray=use_ray)
                m.organize_fluxMatrix(use_ray=use_ray)
                m.get_Tmatrix()
                m.get_steady_state()
                m.get_steady_state_target_flux()
            except Exception as e:  # noqa: BLE001 (msm_we.py:993-1003)
                log.error("Error during block validation!")
                raise modelWE.BlockValidationError(e)
        self.validation_iterations = validation_iterations
        self.validation_models = validation_models

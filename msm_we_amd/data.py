"""Input contract of the hot path (reference ``msm_we/_hamsm/_data.py``).

The reference reads WESTPA HDF5 per iteration through five methods; here the same method names
serve data from an *iteration source* and ``stage_iterations`` moves every iteration once into HBM:

    Xp, Xc  [N, D]      featurized parent / child frames (float32 when exactly representable --
                        WESTPA stores auxdata/coord as float32, augmentation_driver.py:173-180 --
                        else float64)
    p0, p1  [N, ndim]   parent / child progress coordinates, float64
    w       [N]         segment weights, float64
    offsets [maxIter+1] segment offsets of iterations 1..maxIter (host)

A source is any object with ``iterations() -> sorted list[int]`` and ``load(n) -> dict`` holding
``parent`` / ``child`` coordinate arrays ``(nSeg, nAtoms, coord_ndim)`` (or already-featurized
``Xp`` / ``Xc`` ``(nSeg, D)``), ``p0``, ``p1`` ``(nSeg, ndim)`` and ``w`` ``(nSeg,)``.
"""
from __future__ import annotations

import numpy as np

from . import _abi

__all__ = ["DataMixin", "InMemorySource", "NpzDirectorySource", "WestpaH5Source", "DeviceStore"]


class InMemorySource:
    """Iterations held in a dict {n_iter: dict(Xp, Xc, p0, p1, w)} (e.g. from msm_we_amd.synthetic)."""

    def __init__(self, iters: dict):
        self._iters = iters

    def iterations(self):
        return sorted(self._iters)

    def load(self, n):
        return self._iters[int(n)]


class NpzDirectorySource:
    """Iterations as ``<dir>/iter_%08d.npz`` files holding the arrays of the source contract above -- the raw fallback
    for hosts without h5py: export once where the WESTPA run lives (`export`), analyse anywhere.
    Files are memory-mapped lazily, one iteration at a time."""

    PATTERN = "iter_%08d.npz"

    def __init__(self, directory):
        import os
        import re

        self.dir = str(directory)
        found = [re.fullmatch(r"iter_(\d{8})\.npz", f) for f in os.listdir(self.dir)]
        self._its = sorted(int(m.group(1)) for m in found if m)
        if not self._its:
            raise FileNotFoundError(f"no iter_XXXXXXXX.npz files in {self.dir}")

    def iterations(self):
        return list(self._its)

    def load(self, n):
        import os

        with np.load(os.path.join(self.dir, self.PATTERN % int(n))) as z:
            return {k: z[k] for k in z.files}

    @classmethod
    def export(cls, source, directory, iterations=None):
        """write any source (e.g. a WestpaH5Source) in this layout"""
        import os

        os.makedirs(str(directory), exist_ok=True)
        for n in (source.iterations() if iterations is None else iterations):
            np.savez(os.path.join(str(directory), cls.PATTERN % int(n)), **{k: np.asarray(v) for k, v in source.load(n).items()})
        return cls(directory)


class WestpaH5Source:
    """WESTPA ``west.h5`` reader for the layout the reference consumes (``_data.py:853-916, 283-300``):
    ``iterations/iter_%08d/{seg_index['weight'], pcoord, auxdata/<auxpath>}``.  Needs h5py, which is
    an optional dependency (absent in the build image): constructing this class without it raises."""

    def __init__(self, path, auxpath="coord", pcoord_ndim=1):
        try:
            import h5py  # noqa: F401
        except ImportError as e:  # pragma: no cover - h5py is not in the image
            raise ImportError("WestpaH5Source needs h5py") from e
        self.path, self.auxpath, self.ndim = path, auxpath, pcoord_ndim

    def iterations(self):  # pragma: no cover
        import h5py

        with h5py.File(self.path, "r") as f:
            return sorted(int(k.split("_")[1]) for k in f["iterations"].keys())

    def load(self, n):  # pragma: no cover
        import h5py

        with h5py.File(self.path, "r") as f:
            g = f["iterations/iter_%08d" % int(n)]
            pc = g["pcoord"][:]
            co = g["auxdata/" + self.auxpath][:]
            return dict(parent=co[:, 0], child=co[:, -1], p0=pc[:, 0, : self.ndim].astype(np.float64),
                        p1=pc[:, -1, : self.ndim].astype(np.float64), w=g["seg_index"]["weight"].astype(np.float64))


class DeviceStore:
    """All staged iterations, resident in HBM."""

    def __init__(self, h, D, ndim, x_dtype, offsets, Xp, Xc, p0, p1, w):
        self.h, self.D, self.ndim, self.x_dtype = h, D, ndim, x_dtype
        self.offsets = np.asarray(offsets, dtype=np.int64)  # [maxIter+1], iteration it -> [offsets[it-1], offsets[it])
        self.Xp, self.Xc, self.p0, self.p1, self.w = Xp, Xc, p0, p1, w
        self.N = int(self.offsets[-1])
        self.d_offsets = h.to_device(self.offsets)
        # segments with NaN coordinates (`_data.py:303-313, 549-553`): their weight is 0 in the flux matrix (`w_flux`), their
        # iteration is unusable for training, and they may only be discretized inside the basis / target
        self.w_flux = w
        self.nan_parent = self.nan_child = None

    def mark_nan_segments(self, nan_parent, nan_child, w_host):
        if not (nan_parent.any() or nan_child.any()):
            return
        self.nan_parent, self.nan_child = nan_parent, nan_child
        wz = np.array(w_host, dtype=np.float64, copy=True)
        wz[nan_parent | nan_child] = 0.0
        self.w_flux = self.h.to_device(wz)

    @property
    def n_iters(self):
        return len(self.offsets) - 1

    def span(self, it_first, it_last):
        """frame range of iterations it_first..it_last (1-based, inclusive)"""
        return int(self.offsets[it_first - 1]), int(self.offsets[it_last])

    @classmethod
    def from_host(cls, h, Xp, Xc, p0, p1, w, offsets):
        Xp = np.ascontiguousarray(Xp)
        Xc = np.ascontiguousarray(Xc)
        assert Xp.dtype == Xc.dtype and Xp.dtype in (np.float32, np.float64)
        dt = _abi.F32 if Xp.dtype == np.float32 else _abi.F64
        ndim = 1 if np.ndim(p0) == 1 else np.shape(p0)[1]
        p0 = np.ascontiguousarray(p0, dtype=np.float64).reshape(-1, ndim)
        p1 = np.ascontiguousarray(p1, dtype=np.float64).reshape(-1, ndim)
        return cls(h, Xp.shape[1], ndim, dt, offsets, h.to_device(Xp), h.to_device(Xc), h.to_device(p0),
                   h.to_device(p1), h.to_device(np.ascontiguousarray(w, dtype=np.float64)))


class DataMixin:
    seg_weights = {}
    device_featurizer = None  # set by `set_distance_featurizer`

    # -- source plumbing --------------------------------------------------------------------
    def set_iteration_source(self, source):
        self._source = source
        self._store = None

    def _featurize(self, coords):
        """a1 + a2 on the host for now: user ``processCoordinates`` then ``coordinates.transform``
        (docs/usage.rst:41-60; _clustering.py:894-896,1291-1296).  SURVEY §8f-3 moves this on device."""
        X = self.processCoordinates(np.asarray(coords, dtype=np.float64))
        if self.coordinates is not None:
            X = self.coordinates.transform(X)
        return np.ascontiguousarray(X, dtype=np.float64)

    def processCoordinates(self, coords):
        # default featurizer = the notebook's: flatten (examples/hamsm_construction.ipynb cell 9)
        return coords.reshape(coords.shape[0], -1)

    def _load(self, n_iter):
        d = self._source.load(int(n_iter))
        return d

    # -- the reference's loader hooks ---------------------------------------------------------
    def get_iterations(self):
        # `_data.py:934-993`: an iteration counts only if the following one exists (:860-866)
        its = self._source.iterations()
        complete = [i for i in its if (i + 1) in set(its)]
        self.numSegments = np.array([len(self._load(i)["w"]) for i in complete], dtype=float)
        self.maxIter = self.numSegments.size

    def load_iter_data(self, n_iter):
        # `_data.py:807-932`
        d = self._load(n_iter)
        n = len(d["w"])
        self.n_iter = n_iter
        self.westList = np.zeros(n, dtype=int)
        self.segindList = np.arange(n, dtype=int)
        self.seg_weights[n_iter] = np.asarray(d["w"], dtype=np.float64).copy()
        self.weightList = np.asarray(d["w"], dtype=np.float64).copy()
        self.nSeg = n
        self.pcoord0List = np.asarray(d["p0"], dtype=np.float64).reshape(n, -1)[:, : self.pcoord_ndim].copy()
        self.pcoord1List = np.asarray(d["p1"], dtype=np.float64).reshape(n, -1)[:, : self.pcoord_ndim].copy()

    def _coords_of(self, d, which):
        if which in d:
            return np.asarray(d[which], dtype=np.float64)
        X = np.asarray(d["Xp" if which == "parent" else "Xc"], dtype=np.float64)
        return X[:, :, None]  # (nSeg, nAtoms=D, coord_ndim=1)

    def load_iter_coordinates(self):
        # `_data.py:557-618`: segment-END coordinates
        self.cur_iter_coords = self._coords_of(self._load(self.n_iter), "child")

    def load_iter_coordinates0(self):
        self.cur_iter_coords = self._coords_of(self._load(self.n_iter), "parent")

    def get_iter_coordinates(self, iteration):
        # `_data.py:531-555`
        self.load_iter_data(iteration)
        self.load_iter_coordinates()
        bad = np.isnan(self.cur_iter_coords).any(axis=tuple(range(1, self.cur_iter_coords.ndim)))
        return self.cur_iter_coords[~bad]

    def get_transition_data_lag0(self):
        # `_data.py:254-320`
        d = self._load(self.n_iter)
        parent, child = self._coords_of(d, "parent"), self._coords_of(d, "child")
        self.coordPairList = np.stack([parent, child], axis=-1)
        nan_segments = np.isnan(self.coordPairList).any(axis=tuple(range(1, self.coordPairList.ndim)))
        if nan_segments.any():
            self.weightList[nan_segments] = 0.0  # :303-313
        self.transitionWeights = self.weightList.copy()
        self.departureWeights = self.weightList.copy()

    def get_coordSet(self, last_iter, streaming=None, progress_bar=None):
        # `_data.py:677-759` (streaming form: pcoords only) -- and stage everything into HBM
        total = int(sum(self.numSegments[:last_iter]))
        pcoordSet = np.full((total, self.pcoord_ndim), np.nan)
        pos = 0
        for i in range(1, last_iter + 1):
            d = self._load(i)
            n = len(d["w"])
            pcoordSet[pos : pos + n] = np.asarray(d["p1"], dtype=np.float64).reshape(n, -1)[:, : self.pcoord_ndim]
            pos += n
        self.pcoordSet = pcoordSet
        self.first_iter = 1
        self.last_iter = last_iter

    # -- staging ----------------------------------------------------------------------------
    def _linear_map(self):
        """(components [d, Draw], mean [Draw] | None, scale [d] | None) when `self.coordinates` is a fitted
        scikit-learn PCA / IncrementalPCA (`_base.py` transform), else None"""
        c = self.coordinates
        W = getattr(c, "components_", None)
        if W is None or not hasattr(c, "mean_"):
            return None
        scale = None
        if getattr(c, "whiten", False):
            scale = np.sqrt(np.asarray(c.explained_variance_, dtype=np.float64))
            scale[scale < np.finfo(scale.dtype).eps] = np.finfo(scale.dtype).eps
        return np.asarray(W, dtype=np.float64), (None if c.mean_ is None else np.asarray(c.mean_, dtype=np.float64)), scale

    def stage_iterations(self):
        """Featurize and upload iterations 1..maxIter once.  Idempotent.  `processCoordinates` (user code) runs on the
        host; a PCA transform (`reduceCoordinates`, _dimensionality.py:37-66) runs on the device over the uploaded raw
        features; any other `coordinates.transform` runs on the host."""
        if getattr(self, "_store", None) is not None:
            return self._store
        h = self.h
        if getattr(self, "device_featurizer", None) is not None:
            return self._stage_with_device_featurizer()
        lin = self._linear_map()
        parts = {k: [] for k in ("Xp", "Xc", "p0", "p1", "w")}
        sizes = []
        for it in range(1, self.maxIter + 1):
            d = self._load(it)
            n = len(d["w"])
            for which, key in (("parent", "Xp"), ("child", "Xc")):
                if key in d and which not in d:
                    X = np.asarray(d[key])
                    if lin is None and self.coordinates is not None and type(self.coordinates).__name__ != "Coordinates":
                        X = self.coordinates.transform(np.asarray(X, dtype=np.float64))
                elif lin is not None:
                    X = self.processCoordinates(np.asarray(d[which], dtype=np.float64))
                else:
                    X = self._featurize(d[which])
                parts[key].append(X)
            parts["p0"].append(np.asarray(d["p0"], dtype=np.float64).reshape(n, -1)[:, : self.pcoord_ndim])
            parts["p1"].append(np.asarray(d["p1"], dtype=np.float64).reshape(n, -1)[:, : self.pcoord_ndim])
            parts["w"].append(np.asarray(d["w"], dtype=np.float64))
            sizes.append(n)
        Xp = np.concatenate(parts["Xp"])
        Xc = np.concatenate(parts["Xc"])
        # store float32 when that loses nothing (on-disk WESTPA coordinates are float32)
        nan_p, nan_c = np.isnan(Xp).any(axis=1), np.isnan(Xc).any(axis=1)
        if Xp.dtype != np.float32 or Xc.dtype != np.float32:
            Xp32, Xc32 = Xp.astype(np.float32), Xc.astype(np.float32)
            if np.array_equal(Xp32.astype(np.float64), Xp, equal_nan=True) and np.array_equal(Xc32.astype(np.float64), Xc, equal_nan=True):
                Xp, Xc = Xp32, Xc32
            else:
                Xp, Xc = Xp.astype(np.float64), Xc.astype(np.float64)
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        p0, p1, w = np.concatenate(parts["p0"]), np.concatenate(parts["p1"]), np.concatenate(parts["w"])
        if lin is None:
            self._store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offsets)
            self._store.mark_nan_segments(nan_p, nan_c, w)
            return self._store
        W, mean, scale = lin
        dt = _abi.F32 if Xp.dtype == np.float32 else _abi.F64
        n, Draw = Xp.shape
        proj = []
        for X in (Xp, Xc):  # raw features up, reduced features stay: the raw copy is released at once
            d_raw = h.to_device(X)
            proj.append(h.project(d_raw, dt, Draw, n, W, mean, scale))
            d_raw.free()
        self._store = DeviceStore(h, W.shape[0], p0.shape[1], _abi.F64, offsets, proj[0], proj[1], h.to_device(p0), h.to_device(p1),
                                  h.to_device(w))
        self._store.mark_nan_segments(nan_p, nan_c, w)
        return self._store

    def set_distance_featurizer(self, atom_indices, reference_positions):
        """Featurize on the device with the per-atom distance to a reference structure -- what the reference's test
        `processCoordinates` computes with MDAnalysis (tests/test_msm_we.py:19-36).  `atom_indices`: the selection
        (e.g. backbone) into the frame's atoms; `reference_positions`: [(len(atom_indices)), 3] positions of the same
        atoms in the reference structure.  `processCoordinates` is then not called during staging."""
        idx = np.ascontiguousarray(atom_indices, dtype=np.int32)
        ref = np.ascontiguousarray(reference_positions, dtype=np.float32).reshape(len(idx), 3)
        self.device_featurizer = ("dist", idx, ref)
        self._store = None

    def _stage_with_device_featurizer(self):
        """raw coordinates up (float32 when exact), features computed in HBM (`hamsm_featurize_dist`), then the optional
        linear map (`hamsm_project`); the raw copy is released at once"""
        h = self.h
        _, idx, ref = self.device_featurizer
        lin = self._linear_map()
        if lin is None and self.coordinates is not None and type(self.coordinates).__name__ != "Coordinates":
            raise NotImplementedError("device featurizer + a non-linear coordinates.transform")
        raw = {"parent": [], "child": []}
        p0s, p1s, ws, sizes = [], [], [], []
        for it in range(1, self.maxIter + 1):
            d = self._load(it)
            n = len(d["w"])
            for which in raw:
                raw[which].append(np.asarray(d[which]).reshape(n, -1))
            p0s.append(np.asarray(d["p0"], dtype=np.float64).reshape(n, -1)[:, : self.pcoord_ndim])
            p1s.append(np.asarray(d["p1"], dtype=np.float64).reshape(n, -1)[:, : self.pcoord_ndim])
            ws.append(np.asarray(d["w"], dtype=np.float64))
            sizes.append(n)
        feats, nans = [], []
        for which in ("parent", "child"):
            R = np.concatenate(raw[which])
            nans.append(np.isnan(R).any(axis=1))
            R32 = R.astype(np.float32)
            dt, R = (_abi.F32, R32) if np.array_equal(R32.astype(np.float64), R, equal_nan=True) else (_abi.F64, R.astype(np.float64))
            n, width = R.shape
            d_raw = h.to_device(R)
            d_X = h.featurize_dist(d_raw, dt, width, n, width // 3, idx, ref)
            d_raw.free()
            if lin is not None:
                W, mean, scale = lin
                d_Y = h.project(d_X, _abi.F64, len(idx), n, W, mean, scale)
                d_X.free()
                d_X = d_Y
            feats.append(d_X)
        D = len(idx) if lin is None else lin[0].shape[0]
        offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        p0, p1, w = np.concatenate(p0s), np.concatenate(p1s), np.concatenate(ws)
        self._store = DeviceStore(h, D, p0.shape[1], _abi.F64, offsets, feats[0], feats[1], h.to_device(p0), h.to_device(p1),
                                  h.to_device(w))
        self._store.mark_nan_segments(nans[0], nans[1], w)
        return self._store

    def attach_store(self, store: DeviceStore, maxIter=None):
        """Use frames that are already resident (bench.py generates shards straight into HBM)."""
        self._store = store
        if maxIter is not None:
            self.maxIter = maxIter
            self.numSegments = np.diff(store.offsets).astype(float)[:maxIter]

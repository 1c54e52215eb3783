"""Import the *real* reference (``/root/reference``, msm_we 0.1.28.dev1) in the build container so
that golden vectors come from the reference's own code (task ③; recipe: SURVEY.md §8c, App. B).

This file is fixture-generation tooling only: nothing under ``msm_we_amd/`` imports it, and the
tests never import it either (the reference does not exist on the GPU box) -- they read the
``.npz`` files this harness produced.  No reference source is copied: the missing third-party
imports (ray, h5py, mdtraj, westpa, deeptime, scipy.sparse.sputils) are replaced by inert stubs,
and the five HDF5-touching ``DataMixin`` methods are overridden to serve in-memory arrays.
"""
from __future__ import annotations

import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _RemoteFn:
    """ray.remote stand-in: `.remote(*a)` runs synchronously; deliberately not a descriptor, so
    ``self.method.remote(model_id, ...)`` receives exactly the arguments the caller passed."""

    def __init__(self, fn):
        self._fn = fn

    def remote(self, *a, **k):
        return self._fn(*a, **k)

    def __call__(self, *a, **k):
        return self._fn(*a, **k)


class RectilinearBinMapper:
    """Stub of westpa.core.binning.RectilinearBinMapper (third-party, absent here).

    Restated from the reference fixture's pickled state: float32
    boundaries, half-open bins ``b[j] <= p < b[j+1]`` after casting p to float32, row-major
    multi-dim index, uint16 output.  Assumption recorded in DESIGN.md ("parity unpinned" for westpa).
    """

    def __init__(self, boundaries):
        self._boundaries = [np.asarray(b, dtype=np.float32) for b in boundaries]
        self.boundaries = self._boundaries
        self.ndim = len(self._boundaries)
        self.nbins = int(np.prod([len(b) - 1 for b in self._boundaries]))

    def assign(self, coords, mask=None, output=None):
        coords = np.asarray(coords, dtype=np.float32)
        idx = np.zeros(coords.shape[0], dtype=np.int64)
        for d, b in enumerate(self._boundaries):
            j = np.searchsorted(b, coords[:, d], side="right") - 1
            if np.any(j < 0) or np.any(j >= len(b) - 1):
                raise ValueError("coordinate outside bin space")
            idx = idx * (len(b) - 1) + j
        return idx.astype(np.uint16)


class VoronoiBinMapper:
    """Stub of westpa.core.binning.VoronoiBinMapper (third-party, absent here), restated from its documented behaviour:
    float32 coordinates and centres, bin = first argmin of ``dfunc(coordvec, centers, *dfargs, **dfkwargs)``, uint16 output.
    "Parity unpinned" for westpa itself."""

    def __init__(self, dfunc, centers, dfargs=None, dfkwargs=None):
        self.dfunc = dfunc
        self.centers = np.require(centers, dtype=np.float32)
        if self.centers.ndim == 1:
            self.centers = self.centers[:, None]
        self.dfargs, self.dfkwargs = dfargs or (), dfkwargs or {}
        self.ndim = self.centers.shape[1]
        self.nbins = len(self.centers)

    def assign(self, coords, mask=None, output=None):
        coords = np.require(coords, dtype=np.float32).reshape(-1, self.ndim)
        out = np.empty(len(coords), dtype=np.uint16)
        for i, c in enumerate(coords):
            out[i] = np.argmin(self.dfunc(c, self.centers, *self.dfargs, **self.dfkwargs))
        return out


class FuncBinMapper:
    pass


def install_stubs():
    if "ray" in sys.modules and getattr(sys.modules["ray"], "_hamsm_stub", False):
        return
    _mod(
        "ray",
        _hamsm_stub=True,
        remote=lambda fn: _RemoteFn(fn),
        put=lambda x: x,
        get=lambda x: x,
        wait=lambda ids, num_returns=1, timeout=None: (list(ids[:num_returns]), list(ids[num_returns:])),
        is_initialized=lambda: True,
        cluster_resources=lambda: {"CPU": 1},
        available_resources=lambda: {"CPU": 1},
        init=lambda *a, **k: None,
        shutdown=lambda *a, **k: None,
    )
    _mod("h5py", File=object)
    md = _mod("mdtraj")
    md.Trajectory = type("Trajectory", (), {})
    md.Topology = type("Topology", (), {})
    md.load = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("mdtraj stub"))
    w = _mod("westpa")
    w.__path__ = []
    core = _mod("westpa.core")
    core.__path__ = []
    binning = _mod(
        "westpa.core.binning",
        RectilinearBinMapper=RectilinearBinMapper,
        VoronoiBinMapper=VoronoiBinMapper,
        FuncBinMapper=FuncBinMapper,
    )
    core.binning = binning
    w.core = core
    w.analysis = _mod("westpa.analysis", Run=object)
    tools = _mod("westpa.tools")
    tools.__path__ = []
    tools.binning = _mod("westpa.tools.binning")
    w.tools = tools
    w.rc = types.SimpleNamespace(pstatus=print)
    dt = _mod("deeptime")
    dt.__path__ = []
    dt.decomposition = _mod("deeptime.decomposition", TICA=object, VAMP=object)
    import scipy.sparse

    sp = _mod("scipy.sparse.sputils", isdense=lambda x: isinstance(x, np.ndarray))
    scipy.sparse.sputils = sp


class _NpProxy:
    """NumPy >= 1.24 refuses ragged ``np.array`` (reference pins numpy<1.24, environment.yml:6);
    `_clustering.py:1597` builds one.  Fall back to dtype=object exactly there."""

    def __getattr__(self, k):
        return getattr(np, k)

    @staticmethod
    def array(*a, **k):
        try:
            return np.array(*a, **k)
        except ValueError:
            out = np.empty(len(a[0]), dtype=object)
            for i, v in enumerate(a[0]):
                out[i] = v
            return out


class _InlineExecutor:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def submit(self, fn, *a, **k):
        class _R:
            def __init__(s, v=None, e=None):
                s.v, s.e = v, e

            def result(s):
                if s.e is not None:
                    raise s.e
                return s.v

        try:
            return _R(fn(*a, **k))
        except BaseException as e:  # noqa: BLE001 - re-raised by result(), like a real future
            return _R(None, e)


def load_reference(inline_pool=True):
    """Returns the imported reference package.  ``inline_pool`` replaces the fork-per-iteration
    ProcessPoolExecutor (`_clustering.py:672-691`, `_fluxmatrix.py:248-254`) by in-process calls so
    that per-``partial_fit`` state can be recorded; results are byte-identical."""
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import msm_we  # noqa: F401
    import msm_we._hamsm._clustering as cl
    import msm_we._hamsm._fluxmatrix as fm

    cl.np = _NpProxy()
    if inline_pool:
        import msm_we._hamsm._dimensionality as dm

        ns = types.SimpleNamespace(futures=types.SimpleNamespace(ProcessPoolExecutor=_InlineExecutor))
        cl.concurrent = ns
        fm.concurrent = ns
        dm.concurrent = ns  # the fork-per-iteration iPCA loop (_dimensionality.py:176-235)
    return msm_we


def make_model_class(msm_we):
    """modelWE subclass whose HDF5 touchers serve in-memory WE iterations."""

    class InMemoryModelWE(msm_we.modelWE):
        _iters = None  # dict it -> dict(Xp, Xc, p0, p1, w)

        def get_iterations(self):
            its = sorted(self._iters)
            # the reference drops the last (incomplete) iteration: `_data.py:860-866`
            self.numSegments = np.array([self._iters[i]["w"].shape[0] for i in its[:-1]], dtype=float)
            self.maxIter = self.numSegments.size

        def load_iter_data(self, n_iter):
            d = self._iters[int(n_iter)]
            n = d["w"].shape[0]
            self.n_iter = n_iter
            self.westList = np.zeros(n, dtype=int)
            self.segindList = np.arange(n, dtype=int)
            self.seg_weights[n_iter] = d["w"].copy()
            self.weightList = d["w"].copy()
            self.nSeg = n
            self.pcoord0List = d["p0"].copy()
            self.pcoord1List = d["p1"].copy()

        def _coords(self, X):
            # (nSeg, nAtoms=D, coord_ndim=1), float64 like `_data.py:265,565`
            return X.astype(np.float64)[:, :, None]

        def load_iter_coordinates(self):
            self.cur_iter_coords = self._coords(self._iters[int(self.n_iter)]["Xc"])

        def load_iter_coordinates0(self):
            self.cur_iter_coords = self._coords(self._iters[int(self.n_iter)]["Xp"])

        def get_transition_data_lag0(self):
            d = self._iters[int(self.n_iter)]
            cp = np.zeros((self.nSeg, self.nAtoms, self.coord_ndim, 2))
            cp[..., 0] = self._coords(d["Xp"])
            cp[..., 1] = self._coords(d["Xc"])
            self.coordPairList = cp
            # the reference's own method does this after reading HDF5 (`_data.py:303-313`); this override replaces that
            # method, so its NaN rule is restated here: segments with NaN coordinates get weight 0
            nan_segments = np.isnan(cp).any(axis=(1, 2, 3))
            if nan_segments.any():
                self.weightList[nan_segments] = 0.0
            self.transitionWeights = self.weightList.copy()
            self.departureWeights = self.weightList.copy()

    def processCoordinates(self, coords):
        # the notebook featurizer (examples/hamsm_construction.ipynb cell 9): flatten
        return coords.reshape(coords.shape[0], -1)

    InMemoryModelWE.processCoordinates = processCoordinates
    return InMemoryModelWE

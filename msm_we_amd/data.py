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

__all__ = ["DataMixin", "InMemorySource", "NpzDirectorySource", "WestpaH5Source", "DeviceStore", "resolve_object"]


class InMemorySource:
    """Iterations held in a dict {n_iter: dict(Xp, Xc, p0, p1, w)} (e.g. from msm_we_amd.synthetic)."""

    parallel_safe = True   # `load` may be called from several threads at once (staging's reader pool)
    cheap_load = True      # `load` hands out arrays it holds: not worth a thread

    def __init__(self, iters: dict):
        self._iters = iters

    def iterations(self):
        return sorted(self._iters)

    def load(self, n):
        return self._iters[int(n)]


class NpzDirectorySource:
    """Iterations as ``<dir>/iter_%08d.npz`` files holding the arrays of the source contract above -- the raw fallback
    for hosts without h5py: export once where the WESTPA run lives (`export`), analyse anywhere.
    `load` goes through `zipfile` (2 GB/s); staging does not: `np.savez` STORES its members, so every array is a byte range of
    its file (`member_ranges`) that the library's threads read straight into the upload slab."""

    PATTERN = "iter_%08d.npz"
    parallel_safe = True   # every `load` opens its own file

    def __init__(self, directory):
        import os
        import re

        self.dir = str(directory)
        found = [re.fullmatch(r"iter_(\d{8})\.npz", f) for f in os.listdir(self.dir)]
        self._its = sorted(int(m.group(1)) for m in found if m)
        if not self._its:
            raise FileNotFoundError(f"no iter_XXXXXXXX.npz files in {self.dir}")
        self._ranges = {}
        self._load_index()

    INDEX = "member_ranges.json"   # written by `export`: where every array lies in its file (see `member_ranges`)

    def _load_index(self):
        """the byte ranges `export` recorded; an entry is used only while its file still has the recorded size and time stamp"""
        import json
        import os

        try:
            with open(os.path.join(self.dir, self.INDEX)) as f:
                idx = json.load(f)
            for k, e in idx.items():
                path = os.path.join(self.dir, self.PATTERN % int(k))
                st = os.stat(path)
                if st.st_size == e["size"] and st.st_mtime_ns == e["mtime_ns"]:
                    rng = {a: (int(r[0]), int(r[1]), np.dtype(r[2]), tuple(int(v) for v in r[3])) for a, r in e["ranges"].items()}
                    # (the index is a file anyone can edit.  An entry is used only if every range is what its own
                    # shape and dtype say -- the byte count staging will `pread` into the pinned slab -- and lies inside the
                    # file; anything else is dropped and the file is parsed again on first use)
                    if all(o >= 0 and nb == int(np.prod(sh, dtype=np.int64)) * dt.itemsize and o + nb <= st.st_size and not dt.hasobject
                           for o, nb, dt, sh in rng.values()):
                        self._ranges[int(k)] = rng
        except (OSError, ValueError, KeyError, TypeError):
            pass

    def _write_index(self):
        import json
        import os

        idx = {}
        for n in self._its:
            r = self.member_ranges(n)
            if r is not None:
                st = os.stat(os.path.join(self.dir, self.PATTERN % n))
                idx[str(n)] = {"size": st.st_size, "mtime_ns": st.st_mtime_ns,
                               "ranges": {a: [v[0], v[1], v[2].str, list(v[3])] for a, v in r.items()}}
        with open(os.path.join(self.dir, self.INDEX), "w") as f:
            json.dump(idx, f)

    def iterations(self):
        return list(self._its)

    def load(self, n):
        import os

        with np.load(os.path.join(self.dir, self.PATTERN % int(n))) as z:
            return {k: z[k] for k in z.files}

    def member_ranges(self, n):
        """{key: (byte offset, byte count, dtype, shape)} of iteration n's arrays inside its ``.npz``, or None when a member is
        deflated, Fortran-ordered or of object type.  `np.savez` STORES its members: each array's bytes are one contiguous range
        of the file, which staging reads straight into the upload slab (`hamsm_host_read_parts`) instead of going through
        `zipfile` + a copy (2 GB/s).  Parsed once per file (the zip's local headers, then the ``.npy`` header) and kept."""
        import os
        import struct
        import zipfile

        n = int(n)
        cache = self._ranges
        if n in cache:
            return cache[n]
        path = os.path.join(self.dir, self.PATTERN % n)
        out = {}
        try:
            with zipfile.ZipFile(path) as z, open(path, "rb") as f:
                for zi in z.infolist():
                    if zi.compress_type != zipfile.ZIP_STORED or not zi.filename.endswith(".npy"):
                        out = None
                        break
                    f.seek(zi.header_offset)
                    hdr = f.read(30)
                    if len(hdr) != 30 or hdr[:4] != b"PK\x03\x04":
                        out = None
                        break
                    nlen, xlen = struct.unpack("<HH", hdr[26:30])
                    start = zi.header_offset + 30 + nlen + xlen
                    f.seek(start)
                    ver = np.lib.format.read_magic(f)
                    rd = np.lib.format.read_array_header_1_0 if ver == (1, 0) else np.lib.format.read_array_header_2_0
                    shape, fortran, dtype = rd(f)
                    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
                    if (fortran and len(shape) > 1) or dtype.hasobject or f.tell() + nbytes > start + zi.file_size:
                        out = None
                        break
                    out[zi.filename[:-4]] = (int(f.tell()), nbytes, dtype, tuple(int(v) for v in shape))
        except (OSError, ValueError, zipfile.BadZipFile):
            out = None
        cache[n] = out
        return out

    @classmethod
    def export(cls, source, directory, iterations=None):
        """write any source (e.g. a WestpaH5Source) in this layout"""
        import os

        os.makedirs(str(directory), exist_ok=True)
        for n in (source.iterations() if iterations is None else iterations):
            np.savez(os.path.join(str(directory), cls.PATTERN % int(n)), **{k: np.asarray(v) for k, v in source.load(n).items()})
        out = cls(directory)
        out._write_index()
        return out


class NpyDirectorySource:
    """A whole run as raw arrays: ``<dir>/{Xp,Xc,p0,p1,w}.npy`` hold the rows of ALL iterations back to back (iteration-major, like
    the staged frames in HBM), ``offsets.npy`` / ``iterations.npy`` say where each iteration lies.  `load(n)` returns memory-mapped
    slices; staging does not even map them: a block of iterations is one contiguous byte range per array, read with ``preadv``
    by the library's threads STRAIGHT into the page-locked upload slab (`file_range` -> `hamsm_host_read_parts`) -- one copy, no
    page faults, no per-file Python work.  (Per-iteration files read through numpy -- ``.npz`` members or ``.npy`` files --
    measured 1.2-2.0 GB/s on the GPU box whatever the thread count: opening and parsing five arrays per iteration is serial work
    under the interpreter lock.)  The layout to export a long WESTPA run to for repeated analyses."""

    KEYS = ("Xp", "Xc", "p0", "p1", "w")
    parallel_safe = True

    def __init__(self, directory):
        import os

        self.dir = str(directory)
        if not os.path.exists(os.path.join(self.dir, "offsets.npy")):
            raise FileNotFoundError(f"no offsets.npy in {self.dir}")
        self._offsets = np.load(os.path.join(self.dir, "offsets.npy")).astype(np.int64)
        self._its = [int(i) for i in np.load(os.path.join(self.dir, "iterations.npy"))]
        self._index = {it: k for k, it in enumerate(self._its)}
        self._maps = None

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_maps"] = None
        return d

    def _arrays(self):
        import os

        if self._maps is None:
            self._maps = {k: np.load(os.path.join(self.dir, k + ".npy"), mmap_mode="r") for k in self.KEYS}
        return self._maps

    def iterations(self):
        return list(self._its)

    def load(self, n):
        k = self._index[int(n)]
        a, b = int(self._offsets[k]), int(self._offsets[k + 1])
        return {key: arr[a:b] for key, arr in self._arrays().items()}

    # -- the fast path of staging: rows of consecutive iterations, file -> caller's buffer, no intermediate array ---------------
    def row_range(self, it_first, it_last):
        return int(self._offsets[self._index[int(it_first)]]), int(self._offsets[self._index[int(it_last)] + 1])

    def dtype_shape(self, key):
        arr = self._arrays()[key]
        return arr.dtype, arr.shape[1:]

    def file_range(self, key, row0, row1):
        """(path, byte offset, byte count) of rows [row0, row1) of array `key`: what `hamsm_host_read_parts` reads"""
        import os

        arr = self._arrays()[key]
        row_bytes = arr.dtype.itemsize * int(np.prod(arr.shape[1:], dtype=np.int64))
        return os.path.join(self.dir, key + ".npy"), int(arr.offset + row0 * row_bytes), int((row1 - row0) * row_bytes)

    def read_rows(self, key, row0, row1, out):
        """rows [row0, row1) of array `key` into `out` (C-contiguous, same dtype and row shape): ``preadv`` from the file"""
        import os

        arr = self._arrays()[key]
        assert out.dtype == arr.dtype and out.shape == (row1 - row0, *arr.shape[1:]) and out.flags.c_contiguous
        row_bytes = arr.dtype.itemsize * int(np.prod(arr.shape[1:], dtype=np.int64))
        fd = os.open(os.path.join(self.dir, key + ".npy"), os.O_RDONLY)
        try:
            mv = memoryview(out).cast("B")
            done, want, pos = 0, (row1 - row0) * row_bytes, arr.offset + row0 * row_bytes
            while done < want:
                got = os.preadv(fd, [mv[done:]], pos + done)
                if got <= 0:
                    raise IOError(f"{key}.npy ends before row {row1}")
                done += got
        finally:
            os.close(fd)

    @classmethod
    def export(cls, source, directory, iterations=None):
        """write any source (e.g. a WestpaH5Source) in this layout (two passes: sizes, then rows)"""
        import os

        os.makedirs(str(directory), exist_ok=True)
        its = [int(i) for i in (source.iterations() if iterations is None else iterations)]
        first = source.load(its[0])
        names = {"Xp": "Xp" if "Xp" in first else "parent", "Xc": "Xc" if "Xc" in first else "child", "p0": "p0", "p1": "p1", "w": "w"}
        sizes = np.array([len(source.load(i)["w"]) for i in its], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        outs = {}
        for key, src_key in names.items():
            a0 = np.asarray(first[src_key])
            a0 = a0.reshape(len(a0), -1) if key != "w" else a0
            outs[key] = np.lib.format.open_memmap(os.path.join(str(directory), key + ".npy"), mode="w+", dtype=a0.dtype,
                                                  shape=(int(offs[-1]), *a0.shape[1:]))
        for k, it in enumerate(its):
            d_ = source.load(it)
            for key, src_key in names.items():
                a_ = np.asarray(d_[src_key])
                outs[key][offs[k]: offs[k + 1]] = a_.reshape(len(a_), -1) if key != "w" else a_
        for o in outs.values():
            o.flush()
        del outs
        np.save(os.path.join(str(directory), "offsets.npy"), offs)
        np.save(os.path.join(str(directory), "iterations.npy"), np.array(its, dtype=np.int64))
        return cls(directory)


def resolve_object(spec):
    """``westpa.core.extloader.get_object``: 'package.module.attribute' -> object (objects pass through; westpa's own
    loader is used where WESTPA is installed)."""
    if not isinstance(spec, str):
        return spec
    try:
        from westpa.core import extloader

        return extloader.get_object(spec)
    except ImportError:
        import importlib

        module, _, attr = spec.rpartition(".")
        if not module:
            raise ValueError(f"{spec!r} is not of the form package.module.attribute")
        return getattr(importlib.import_module(module), attr)


class WestpaH5Source:
    """WESTPA ``west.h5`` reader for the layout the reference consumes (``_data.py:853-916, 283-300``; written by
    ``augmentation_driver.py:173-180``): ``iterations/iter_%08d/{seg_index['weight'], pcoord (nSeg, pcoord_len, ndim),
    auxdata/<auxpath> (nSeg, 2, nAtoms, 3) float32}``.  Several files (the reference's ``fileList``) are read one after the
    other, an iteration's segments concatenated file by file like ``load_iter_data`` does; ``west`` / ``segind`` carry every
    segment's file index and its index inside that file (the reference's ``westList`` / ``segindList``, ``_data.py:893-895``).

    Files are open only while somebody reads: a single ``load`` / ``iterations`` call opens and closes them like the reference
    does (``_data.py:853, 918``), and a pass over many iterations wraps itself in ``with source.session():`` so that the files are
    opened once per pass, not per iteration (``DataMixin`` does this for ``get_iterations``, ``get_coordSet``, ``dimReduce`` and
    ``stage_iterations``).  No handle outlives the pass: WESTPA and the restart driver can reopen ``west.h5`` for writing, and
    the source -- and a model that holds it -- pickles (paths only).
    Needs h5py, an optional dependency (absent in the build image: the tests run this class against a stand-in module)."""

    def __init__(self, path, auxpath="coord", pcoord_ndim=1):
        self._h5py()
        self.paths = [path] if isinstance(path, (str, bytes)) or hasattr(path, "__fspath__") else list(path)
        self.auxpath, self.ndim = auxpath, pcoord_ndim
        self._files, self._depth = None, 0

    @staticmethod
    def _h5py():
        try:
            import h5py
        except ImportError as e:
            raise ImportError("WestpaH5Source needs h5py") from e
        return h5py

    def session(self):
        """context manager: the files stay open until the outermost session ends"""
        return _H5Session(self)

    def close(self):
        for f in self._files or ():
            f.close()
        self._files = None

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_files"], d["_depth"] = None, 0
        return d

    def iterations(self):
        with self.session():
            its = None
            for f in self._files:
                here = {int(k.split("_")[1]) for k in f["iterations"].keys()}
                its = here if its is None else its | here
            return sorted(its or ())

    def load(self, n):
        key = "iterations/iter_%08d" % int(n)
        nxt = "iterations/iter_%08d/seg_index" % (int(n) + 1)
        parts = []
        with self.session():
            for file_idx, f in enumerate(self._files):
                # the reference's test is PER FILE (`_data.py:857-866`): a file contributes iteration n only if it also holds
                # iteration n + 1 ("last iterations have incomplete data") -- with files of different lengths the segment sets,
                # westList / segindList and numSegments depend on it
                if (key + "/seg_index") not in f or nxt not in f:
                    continue
                g = f[key]
                pc = np.asarray(g["pcoord"][:])
                co = np.asarray(g["auxdata/" + self.auxpath][:])
                parts.append(dict(parent=co[:, 0], child=co[:, -1], p0=pc[:, 0, : self.ndim].astype(np.float64),
                                  p1=pc[:, -1, : self.ndim].astype(np.float64),
                                  w=np.asarray(g["seg_index"]["weight"], dtype=np.float64),
                                  west=np.full(len(pc), file_idx, dtype=int), segind=np.arange(len(pc), dtype=int)))
        if not parts:
            raise KeyError(key)
        return parts[0] if len(parts) == 1 else {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}


class _H5Session:
    def __init__(self, src):
        self.src = src

    def __enter__(self):
        s = self.src
        if s._files is None:
            h5py = s._h5py()
            s._files = [h5py.File(p, "r") for p in s.paths]
        s._depth += 1
        return s

    def __exit__(self, *exc):
        s = self.src
        s._depth -= 1
        if s._depth <= 0:
            s._depth = 0
            s.close()
        return False


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

    def mark_nan_segments(self, nan_parent, nan_child, w_host=None):
        """`w_host`: the weights on the host if the caller holds them; read back from HBM otherwise (only when a NaN row exists)"""
        if not (nan_parent.any() or nan_child.any()):
            return
        self.nan_parent, self.nan_child = nan_parent, nan_child
        wz = np.array(self.w.to_host() if w_host is None else w_host, dtype=np.float64, copy=True).reshape(-1)
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
        self._train_store = None

    def _featurize(self, coords):
        """a1 + a2 on the host for now: user ``processCoordinates`` then ``coordinates.transform``
        (docs/usage.rst:41-60; _clustering.py:894-896,1291-1296).  SURVEY §8f-3 moves this on device."""
        X = self.processCoordinates(np.asarray(coords, dtype=np.float64))
        if self.coordinates is not None:
            X = self.coordinates.transform(X)
        return np.ascontiguousarray(X, dtype=np.float64)

    # -- a1: the user's featurizer.  The reference's idiom is a class patch, `modelWE.processCoordinates = f`
    #    (docs/usage.rst:41-60; hamsm_driver.py:100-103) -- that still works here (or a subclass).  `set_featurizer` binds one to
    #    THIS model instead, without a class of its own, so that the model stays picklable (restart_driver.py:1142 pickles the
    #    plugin's model): a dotted path is what the pickle carries, resolved again on first use after loading. ---------------------
    _featurizer = None
    _featurizer_spec = None

    def set_featurizer(self, featurizer):
        """`featurizer(model, coords) -> (n, D)`: a callable, or its 'package.module.function' path (west.cfg's
        `featurization`)"""
        self._featurizer_spec = featurizer if isinstance(featurizer, str) else None
        self._featurizer = resolve_object(featurizer)
        self._store = None
        self._train_store = None

    def processCoordinates(self, coords):
        f = self._featurizer
        if f is None and self._featurizer_spec is not None:
            f = self._featurizer = resolve_object(self._featurizer_spec)
        if f is not None:
            return f(self, coords)
        # default featurizer = the notebook's: flatten (examples/hamsm_construction.ipynb cell 9)
        return coords.reshape(coords.shape[0], -1)

    def _load(self, n_iter):
        d = self._source.load(int(n_iter))
        return d

    def _source_session(self):
        """a pass over many iterations: sources that hold files (WestpaH5Source) keep them open for its duration only"""
        import contextlib

        sess = getattr(getattr(self, "_source", None), "session", None)
        return sess() if callable(sess) else contextlib.nullcontext()

    # -- the reference's loader hooks ---------------------------------------------------------
    def get_iterations(self):
        # `_data.py:934-993`: an iteration counts only if the following one exists (:860-866)
        with self._source_session():
            its = self._source.iterations()
            complete = [i for i in its if (i + 1) in set(its)]
            segs = []
            for i in complete:   # (the reference counts on while an iteration has segments, :953-987; a source whose files
                try:             # hold an iteration but not its successor in the SAME file yields none: KeyError)
                    segs.append(len(self._load(i)["w"]))
                except KeyError:
                    break
            self.numSegments = np.array(segs, dtype=float)
        self.maxIter = self.numSegments.size

    def load_iter_data(self, n_iter):
        # `_data.py:807-932`
        d = self._load(n_iter)
        n = len(d["w"])
        self.n_iter = n_iter
        # provenance of every segment (`_data.py:893-895`): index into `fileList`, index inside that file
        self.westList = np.asarray(d["west"], dtype=int) if "west" in d else np.zeros(n, dtype=int)
        self.segindList = np.asarray(d["segind"], dtype=int) if "segind" in d else np.arange(n, dtype=int)
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
        with self._source_session():
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

    # -- staging: one pass over the iterations, host work overlapped with the uploads --------------
    STAGE_CHUNK_BYTES = 32 << 20  # host bytes per upload block (page-locking + releasing the two slabs costs ~0.3 ms per MB: DESIGN.md 5.5)
    STAGE_PIECE_BYTES = 2 << 20   # least bytes per reader job
    STAGE_JOBS_PER_READER = 1     # a block is cut into at most readers x this many jobs

    def _iteration_host(self, it, mode):
        """host side of one iteration: (Xp, Xc, p0, p1, w) ready for the upload slab"""
        d = self._load(it)
        n = len(d["w"])
        X = []
        for which, key in (("parent", "Xp"), ("child", "Xc")):
            if mode == "device_featurizer":
                x = np.asarray(d[which]).reshape(n, -1)           # raw coordinates, featurized in HBM
            elif key in d and which not in d:
                x = np.asarray(d[key])                            # already featurized by the source
                if mode == "host" and self.coordinates is not None and type(self.coordinates).__name__ != "Coordinates":
                    x = self.coordinates.transform(np.asarray(x, dtype=np.float64))
            elif mode == "device_pca":
                x = self.processCoordinates(np.asarray(d[which], dtype=np.float64))   # raw features, projected in HBM
            else:
                x = self._featurize(d[which])
            X.append(x)
        return (X[0], X[1], np.asarray(d["p0"], dtype=np.float64).reshape(n, -1)[:, : self.pcoord_ndim],
                np.asarray(d["p1"], dtype=np.float64).reshape(n, -1)[:, : self.pcoord_ndim], np.asarray(d["w"], dtype=np.float64))

    def _features_are_stored(self):
        """the staged features ARE the source's `Xp` / `Xc` arrays: built-in flattening featurizer, no host-side transform"""
        builtin = type(self).processCoordinates is DataMixin.processCoordinates and self._featurizer is None and self._featurizer_spec is None
        ident = self.coordinates is None or type(self.coordinates).__name__ == "Coordinates"
        return builtin and ident and getattr(self, "device_featurizer", None) is None

    STAGE_READERS = None  # reader threads per block; None: up to 12 where the source and the featurizer allow it, else 1

    def _stage_readers(self):
        """how many threads may load / featurize iterations at once: the source must say it is safe (`parallel_safe`: the in-memory
        and per-iteration-file sources; not the HDF5 one, h5py handles are not) and `processCoordinates` must be the built-in
        flattening (user code is called from one thread, as the reference calls it)"""
        if self.STAGE_READERS is not None:
            return max(1, int(self.STAGE_READERS))
        import os

        builtin = type(self).processCoordinates is DataMixin.processCoordinates and self._featurizer is None and self._featurizer_spec is None
        if not (getattr(self._source, "parallel_safe", False) and builtin):
            return 1
        return max(1, min(12, (os.cpu_count() or 2) - 1))

    def _fill_slab(self, its, mode, slab, up_dtype, preloaded=None, pool=None):
        """reader thread: iterations `its` written straight into a page-locked slab (one copy, no intermediate concatenation);
        with a `pool`, several iterations are loaded at once, each into its own rows (their counts are known beforehand).
        `preloaded`: {iteration: its host block} for blocks that are in memory already.
        -> (rows, nan_parent, nan_child, fits): `fits` is False when a value does not survive the slab's dtype"""

        PIECE = int(self.STAGE_PIECE_BYTES)   # bytes per copy job: small enough to spread a block over the readers, large enough to hide the dispatch
        nt = max(1, getattr(pool, "_max_workers", 1) if pool is not None else 1)
        run = (lambda f, jobs: list(pool.map(f, jobs))) if pool is not None else (lambda f, jobs: [f(j) for j in jobs])
        row_bytes = 2 * slab["X"][0].shape[1] * slab["X"][0].dtype.itemsize + 16 * slab["p0"].shape[1] + 8

        def cuts_of(n):
            k = int(max(1, min(nt * int(self.STAGE_JOBS_PER_READER), (n * row_bytes) // PIECE)))
            c = np.linspace(0, n, k + 1).astype(np.int64)
            return [(int(c[t]), int(c[t + 1])) for t in range(k) if c[t + 1] > c[t]]

        sizes = getattr(self, "_stage_sizes", None)
        src = self._source
        if (mode == "host" and hasattr(src, "read_rows") and self._features_are_stored()
                and list(its) == list(range(its[0], its[-1] + 1))
                and all(src.dtype_shape(k) == (np.dtype(up_dtype), slab["X"][0].shape[1:]) for k in ("Xp", "Xc"))
                and all(src.dtype_shape(k) == (np.dtype(np.float64), slab["p0"].shape[1:]) for k in ("p0", "p1"))
                and src.dtype_shape("w") == (np.dtype(np.float64), ())):
            # a block of consecutive iterations is ONE byte range per stored array: its parts go file -> slab, in parallel
            r0, r1 = src.row_range(its[0], its[-1])
            n = r1 - r0
            if sizes is not None and n != sum(sizes[it] for it in its):
                raise ValueError(f"iterations {its[0]}..{its[-1]} hold {n} segments, another number was announced")
            if n > len(slab["w"]):
                raise ValueError(f"iterations {its[0]}..{its[-1]} hold {n} segments, the upload slab was sized for {len(slab['w'])}")
            dsts = (("Xp", slab["X"][0]), ("Xc", slab["X"][1]), ("p0", slab["p0"]), ("p1", slab["p1"]), ("w", slab["w"]))

            def rd(ab):
                for key, dst in dsts:
                    src.read_rows(key, r0 + ab[0], r0 + ab[1], dst[ab[0]: ab[1]])

            if hasattr(src, "file_range") and n > 0:
                # the library's own threads pread the five ranges (no interpreter lock, pieces of 1 MB dealt out by a counter)
                import os

                fds = []
                try:
                    offs_, nb_ = [], []
                    for key, dst in dsts:
                        path, fo, nb = src.file_range(key, r0, r1)
                        if nb != dst[:n].nbytes or not dst.flags.c_contiguous:
                            raise ValueError(f"{key}: {nb} bytes on file for {n} rows of a slab that holds {dst[:n].nbytes}")
                        fds.append(os.open(path, os.O_RDONLY))
                        offs_.append(fo)
                        nb_.append(nb)
                    self.h.host_read_parts(fds, offs_, [d_.__array_interface__["data"][0] for _, d_ in dsts], nb_, threads=nt)
                finally:
                    for fd in fds:
                        os.close(fd)
            else:
                run(rd, cuts_of(n))
            self.__dict__["_stage_direct_blocks"] = self.__dict__.get("_stage_direct_blocks", 0) + 1
            return n, None, None, True

        if mode == "host" and hasattr(src, "member_ranges") and hasattr(src, "PATTERN") and self._features_are_stored() and len(its):
            # per-iteration ``.npz`` files whose members are stored: every array is a byte range of its file -> file to slab by the
            # library's threads, no zipfile, no intermediate arrays.  (Anything unusual -- a deflated member, another dtype or
            # width, raw coordinates instead of features -- takes the general path below.)
            import os

            dsts = (("Xp", slab["X"][0]), ("Xc", slab["X"][1]), ("p0", slab["p0"]), ("p1", slab["p1"]), ("w", slab["w"]))
            rngs = [src.member_ranges(it) for it in its]
            plain = all(r is not None and all(k in r and r[k][2] == d_.dtype and r[k][3][1:] == d_.shape[1:] and r[k][3][:1] == r["w"][3][:1]
                                              for k, d_ in dsts) for r in rngs)
            if plain:
                ns = [int(r["w"][3][0]) for r in rngs]
                if sizes is not None and len(its) > 1:
                    for it, n_ in zip(its, ns):
                        if n_ != sizes[it]:
                            raise ValueError(f"iteration {it} holds {n_} segments, {sizes[it]} were announced")
                n_all = int(sum(ns))
                if n_all > len(slab["w"]):
                    raise ValueError(f"iterations {its[0]}..{its[-1]} hold {n_all} segments, the upload slab was sized for {len(slab['w'])}")
                fds, f_of, offs_, dp, nb_ = [], [], [], [], []
                try:
                    o_ = 0
                    for it, r, n_ in zip(its, rngs, ns):
                        fds.append(os.open(os.path.join(src.dir, src.PATTERN % int(it)), os.O_RDONLY))
                        for k, d_ in dsts:
                            if r[k][1] != n_ * (d_.nbytes // max(1, len(d_))):   # (what the slab has room for, row for row)
                                raise ValueError(f"iteration {it}: member {k} holds {r[k][1]} bytes, {n_} rows of the slab's width were expected")
                            if r[k][1]:
                                f_of.append(fds[-1])
                                offs_.append(r[k][0])
                                dp.append(d_.__array_interface__["data"][0] + o_ * (d_.nbytes // max(1, len(d_))))
                                nb_.append(r[k][1])
                        o_ += n_
                    if nb_:
                        self.h.host_read_parts(f_of, offs_, dp, nb_, threads=nt)
                finally:
                    for fd in fds:
                        os.close(fd)
                self.__dict__["_stage_direct_blocks"] = self.__dict__.get("_stage_direct_blocks", 0) + 1
                return n_all, None, None, True

        # 1. the iterations' host blocks (a file read, the featurizer: parallel by iteration; a source that only hands out arrays
        # it holds is asked in this thread -- a pool dispatch costs ~60 us, more than the lookup); 2. the block's rows into the
        # slab in pieces of >= PIECE bytes whatever the iterations' sizes (a block of one 27 MB iteration still uses every reader,
        # a block of fifty small ones is not fifty dispatches)
        load = lambda it: preloaded[it] if preloaded and it in preloaded else self._iteration_host(it, mode)  # noqa: E731
        cheap = getattr(src, "cheap_load", False) and (mode == "device_featurizer" or self._features_are_stored())
        blks = [load(it) for it in its] if cheap else run(load, list(its))
        starts = np.concatenate([[0], np.cumsum([len(b_[4]) for b_ in blks])]).astype(np.int64)
        if sizes is not None and len(its) > 1:
            for it, b_ in zip(its, blks):
                if len(b_[4]) != sizes[it]:
                    raise ValueError(f"iteration {it} holds {len(b_[4])} segments, {sizes[it]} were announced")
        n_all = int(starts[-1])
        if n_all > len(slab["w"]):   # (the copies below write raw memory: never past the slab)
            raise ValueError(f"iterations {its[0]}..{its[-1]} hold {n_all} segments, the upload slab was sized for {len(slab['w'])}")

        def put(ab):
            fits = True
            for q in range(int(np.searchsorted(starts, ab[0], side="right")) - 1, len(blks)):
                o_ = int(starts[q])
                a, b = max(ab[0], o_) - o_, min(ab[1], int(starts[q + 1])) - o_
                if b <= a:
                    if o_ >= ab[1]:
                        break
                    continue
                blk = blks[q]
                for side in range(2):
                    x = blk[side][a:b]
                    dst = slab["X"][side][o_ + a: o_ + b]
                    dst[...] = x                                  # (casts when the dtypes differ)
                    # (NaN-coordinate segments are found on the device, where the block lands: `hamsm_nan_rows` -- a host scan
                    # runs at 3 GB/s per core, slower than this copy)
                    if x.dtype != up_dtype and up_dtype == np.float32 and fits:
                        fits = bool(np.array_equal(dst.astype(x.dtype), x, equal_nan=True))
                slab["p0"][o_ + a: o_ + b], slab["p1"][o_ + a: o_ + b], slab["w"][o_ + a: o_ + b] = blk[2][a:b], blk[3][a:b], blk[4][a:b]
            return fits

        want = (np.dtype(up_dtype), np.dtype(up_dtype), np.dtype(np.float64), np.dtype(np.float64), np.dtype(np.float64))
        if all(isinstance(a_, np.ndarray) and a_.dtype == dt_ and a_.flags.c_contiguous for b_ in blks for a_, dt_ in zip(b_, want)):
            # same bytes on both sides: the library's threads memcpy them (no cast, so nothing to check: `fits` holds)
            dsts = (slab["X"][0], slab["X"][1], slab["p0"], slab["p1"], slab["w"])
            sp, dp, nb = [], [], []
            for q, b_ in enumerate(blks):
                o_ = int(starts[q])
                for a_, d_ in zip(b_, dsts):
                    if a_.nbytes:
                        if a_.shape[1:] != d_.shape[1:] or len(a_) != len(b_[4]):
                            raise ValueError(f"iteration {its[q]}: an array of shape {a_.shape} for {len(b_[4])} segments of row shape {d_.shape[1:]}")
                        sp.append(a_.__array_interface__["data"][0])
                        dp.append(d_.__array_interface__["data"][0] + o_ * (d_.nbytes // max(1, len(d_))))
                        nb.append(a_.nbytes)
            if nb:
                self.h.host_copy_parts(sp, dp, nb, threads=nt)
            return n_all, None, None, True
        return n_all, None, None, all(run(put, cuts_of(n_all)))

    def stage_iterations(self):
        """Featurize and upload iterations 1..maxIter once.  Idempotent.  ONE pass over the source: a reader thread loads and
        featurizes block k+1 (`processCoordinates` is user code and stays on the host) straight into one of two page-locked
        slabs while block k travels from the other on the context's copy stream -- and while the device turns the uploaded
        block into features where that is its job: the shipped distance featurizer (`hamsm_featurize_dist`) and / or a fitted
        PCA's transform (`hamsm_project`; `reduceCoordinates`, _dimensionality.py:37-66).  Any other `coordinates.transform`
        runs on the host.  Frames are stored float32 when that loses nothing (WESTPA's auxdata/coord is float32), else
        float64: the first iteration decides how the pass starts, and if a later block holds a value float32 cannot carry the
        pass switches in place -- what is in HBM already is widened there (`hamsm_widen_f32`, exact) and that one block is
        filled again; nothing else is read twice."""
        if getattr(self, "_store", None) is not None:
            return self._store
        import time
        from concurrent.futures import ThreadPoolExecutor

        h = self.h
        t_start = time.perf_counter()
        phase = {"first_iteration": 0.0, "alloc": 0.0, "wait_fill": 0.0, "wait_copy": 0.0, "release": 0.0, "finish": 0.0}   # where the host thread stood still

        def lap(key, t0):
            phase[key] += time.perf_counter() - t0

        lin = self._linear_map()
        feat = getattr(self, "device_featurizer", None)
        if feat is not None and lin is None and self.coordinates is not None and type(self.coordinates).__name__ != "Coordinates":
            raise NotImplementedError("device featurizer + a non-linear coordinates.transform")
        mode = "device_featurizer" if feat is not None else ("device_pca" if lin is not None else "host")
        its = list(range(1, self.maxIter + 1))
        if not its:
            raise ValueError("no complete iteration to stage (get_iterations found none)")
        with self._source_session():
            sizes = np.array([len(self._load(i)["w"]) for i in its], dtype=np.int64) if len(getattr(self, "numSegments", ())) != len(its) \
                else np.asarray(self.numSegments, dtype=np.int64)
            offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            N = int(offsets[-1])
            t0_ = time.perf_counter()
            first = self._iteration_host(its[0], mode)     # fixes the widths and the dtype the pass starts in; reused as block 0
            lap("first_iteration", t0_)
            preloaded = {its[0]: first}
            raw_w, ndim = first[0].shape[1], first[2].shape[1]
            row_bytes = 2 * raw_w * first[0].dtype.itemsize + 16 * ndim + 8
            per_chunk = max(1, int(self.STAGE_CHUNK_BYTES // max(1, row_bytes * max(1, int(sizes.mean())))))
            chunks = [its[i: i + per_chunk] for i in range(0, len(its), per_chunk)]
            rows_max = int(max(offsets[c[-1]] - offsets[c[0] - 1] for c in chunks))
            f32_exact = lambda x: x.dtype == np.float32 or bool(  # noqa: E731
                np.array_equal(np.asarray(x, dtype=np.float32).astype(x.dtype), x, equal_nan=True))
            up_dtype = np.float32 if all(f32_exact(np.asarray(x)) for x in first[:2]) else np.float64
            D = raw_w if mode == "host" else ((len(feat[1]) if feat is not None else raw_w) if lin is None else lin[0].shape[0])
            n_all = max(N, 1)

            def alloc_slab(dt):
                return dict(X=[h.pinned((rows_max, raw_w), dt) for _ in range(2)], p0=h.pinned((rows_max, ndim), np.float64),
                            p1=h.pinned((rows_max, ndim), np.float64), w=h.pinned(rows_max, np.float64))

            def alloc_slabs(dt):
                return [alloc_slab(dt) for _ in range(2)]

            def free_slabs(slabs_):
                for sl_ in slabs_:
                    for a_ in (*sl_["X"], sl_["p0"], sl_["p1"], sl_["w"]):
                        h.free_pinned(a_)

            raw = mid = None
            Xd, d_rest = [], []
            d_p0 = d_p1 = d_w = d_nan = None
            t0_ = time.perf_counter()
            slabs = [alloc_slab(up_dtype)]               # (the second slab and the device arrays: while block 0 is being read)
            lap("alloc", t0_)
            scan = None                                  # host mode: the block whose upload is in flight (scanned once it has landed)

            def scan_now():
                if scan is not None:
                    f0_, n_, dt_ = scan
                    for side_ in range(2):
                        h.nan_rows(Xd[side_].offset(f0_ * D), dt_, D, D, n_, d_nan.offset(side_ * max(N, 1) + f0_))

            widened = False
            done = False
            n_readers = self._stage_readers()
            readers = ThreadPoolExecutor(n_readers) if n_readers > 1 else None
            self._stage_sizes = {it: int(sizes[k]) for k, it in enumerate(its)}
            self.__dict__["_stage_direct_blocks"] = 0
            try:
                with ThreadPoolExecutor(1) as pool:
                    fut = pool.submit(self._fill_slab, chunks[0], mode, slabs[0], up_dtype, preloaded, readers)
                    # page-locking the second slab and allocating the device arrays (20-30 ms per GB staged) overlap the reading of
                    # block 0
                    t0_ = time.perf_counter()
                    Xd = [h.alloc(n_all * D, up_dtype if mode == "host" else np.float64) for _ in range(2)]
                    if mode != "host":
                        raw = [h.alloc(max(rows_max, 1) * raw_w, up_dtype) for _ in range(2)]       # uploaded block, parent / child
                        if feat is not None and lin is not None:
                            mid = [h.alloc(max(rows_max, 1) * len(feat[1]), np.float64) for _ in range(2)]
                    d_p0, d_p1, d_w = h.alloc(n_all * ndim, np.float64), h.alloc(n_all * ndim, np.float64), h.alloc(n_all, np.float64)
                    d_rest = [d_p0, d_p1, d_w]
                    slabs.append(alloc_slab(up_dtype))
                    d_nan = h.zeros((2, max(N, 1)), np.uint8)   # rows with a NaN, parent / child: filled block by block on the device
                    d_rest.append(d_nan)
                    lap("alloc", t0_)
                    for k, ch in enumerate(chunks):
                        t0_ = time.perf_counter()
                        n, _, _, fits = fut.result()                  # slab k & 1 holds block k
                        lap("wait_fill", t0_)
                        t0_ = time.perf_counter()
                        h.copy_sync()                                 # block k-1 has left the other slab ...
                        lap("wait_copy", t0_)
                        scan_now()                                    # ... and lies in HBM: its NaN rows (host mode)
                        scan = None
                        if mode != "host":
                            h.sync()                                  # ... and the kernels that read its raw upload are done
                        if not fits:
                            # float32 cannot carry this block: the rest of the pass runs in float64.  Blocks 0..k-1 are exact in
                            # float32 and are widened where they lie; block k is filled again (its slab held rounded values)
                            f0 = int(offsets[ch[0] - 1])
                            h.sync()
                            if mode == "host":
                                for side in range(2):
                                    wide = h.alloc(n_all * D, np.float64)
                                    if f0:
                                        h.widen_f32(Xd[side], wide, f0 * D)
                                    h.sync()
                                    Xd[side].free()
                                    Xd[side] = wide
                            else:
                                for b_ in raw:
                                    b_.free()
                                raw = []
                                raw = [h.alloc(max(rows_max, 1) * raw_w, np.float64) for _ in range(2)]
                            free_slabs(slabs)
                            slabs = []   # (if the allocation below raises, the `finally` must not meet the freed slabs again)
                            up_dtype, widened = np.float64, True
                            slabs = alloc_slabs(up_dtype)
                            n, _, _, fits = self._fill_slab(ch, mode, slabs[k & 1], up_dtype, preloaded, readers)
                        if k + 1 < len(chunks):
                            fut = pool.submit(self._fill_slab, chunks[k + 1], mode, slabs[(k + 1) & 1], up_dtype, None, readers)
                        up_dt = _abi.F32 if up_dtype == np.float32 else _abi.F64
                        f0 = int(offsets[ch[0] - 1])
                        assert n == int(offsets[ch[-1]]) - f0
                        if n == 0:
                            continue
                        sl = slabs[k & 1]
                        for side in range(2):
                            dst = Xd[side].offset(f0 * D) if mode == "host" else raw[side]
                            h.h2d_async(dst, sl["X"][side], n * raw_w * np.dtype(up_dtype).itemsize)
                        h.h2d_async(d_p0.offset(f0 * ndim), sl["p0"], n * ndim * 8)
                        h.h2d_async(d_p1.offset(f0 * ndim), sl["p1"], n * ndim * 8)
                        h.h2d_async(d_w.offset(f0), sl["w"], n * 8)
                        if mode == "host":
                            scan = (f0, n, up_dt)
                        else:
                            h.copy_sync()   # the device stage reads the uploaded block
                            for side in range(2):
                                h.nan_rows(raw[side], up_dt, raw_w, raw_w, n, d_nan.offset(side * max(N, 1) + f0))
                                src, src_dt, src_w = raw[side], up_dt, raw_w
                                if feat is not None:
                                    out = mid[side] if lin is not None else Xd[side].offset(f0 * D)
                                    h.featurize_dist(src, src_dt, src_w, n, src_w // 3, feat[1], feat[2], d_X=out)
                                    src, src_dt, src_w = out, _abi.F64, len(feat[1])
                                if lin is not None:
                                    h.project(src, src_dt, src_w, n, lin[0], lin[1], lin[2], d_Y=Xd[side].offset(f0 * D))
                    h.copy_sync()
                    scan_now()
                    h.sync()
                    done = True
            finally:
                t0_ = time.perf_counter()
                if readers is not None:
                    readers.shutdown(wait=True)
                self._stage_sizes = None
                free_slabs(slabs)
                for bufs in (raw, mid):
                    for b_ in bufs or ():
                        b_.free()
                lap("release", t0_)
                if not done:
                    for b_ in (*Xd, *d_rest):
                        b_.free()
        t0_ = time.perf_counter()
        x_dt = (_abi.F32 if up_dtype == np.float32 else _abi.F64) if mode == "host" else _abi.F64
        store = DeviceStore(h, D, ndim, x_dt, offsets, Xd[0], Xd[1], d_p0, d_p1, d_w)
        nan = d_nan.to_host().astype(bool)[:, :N]
        d_nan.free()
        store.mark_nan_segments(nan[0], nan[1])
        self._store = store
        lap("finish", t0_)
        nbytes = N * (2 * raw_w * (4 if store.x_dtype == _abi.F32 and mode == "host" else first[0].dtype.itemsize) + 16 * ndim + 8)
        self.ingest_stats = {"seconds": time.perf_counter() - t_start, "bytes": int(nbytes), "blocks": len(chunks), "mode": mode,
                             "widened_mid_pass": widened, "readers": n_readers, "direct_read_blocks": self.__dict__.get("_stage_direct_blocks", 0), "host_waits_s": {k: round(v, 5) for k, v in phase.items()}}
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

    # -- sharded runs: the frames training needs, on the rank that trains --------------------------------------------
    train_shards = "all"   # "all": rank 0 trains on every rank's iterations (N ranks == one process on the same data);
                           # "strata": the same model, but the strata's independent chains are dealt over the ranks (stratum s
                           #   trains on rank s % N, which is sent the rows of its strata and nothing else);
                           # "rank0": on its own shard only (the reference's `iters_to_use` restricted to that block)

    def training_store(self):
        """(store, maxIter) the training pass runs over.  One rank, or `train_shards == "rank0"`: the local store.  Otherwise
        the child frames, parent pcoords and weights of iterations 1..maxIter-1 of EVERY rank are brought to rank 0 once
        (`hamsm_gather_to_root`: point-to-point over xGMI; they stay resident next to rank 0's own shard) and chained in rank
        order -- shards are consecutive blocks of WE iterations -- so that training sees exactly the frames, in exactly the
        order, a single process holding all iterations would.  Other ranks get (None, 0): they do not train."""
        h = self.h
        R = getattr(h, "nranks", 1)
        store = self.stage_iterations()
        if R == 1 or self.train_shards != "all":
            # ("strata": every rank is sent the rows of its own strata by `ClusteringMixin._strata_training_store`, which needs
            # the training plan -- cluster count, iterations -- and therefore runs inside `cluster_stratified`)
            return store, self.maxIter
        if getattr(self, "_train_store", None) is not None:
            return self._train_store
        if store.nan_child is not None:
            raise NotImplementedError("NaN-coordinate segments in a sharded run with train_shards='all'")
        n_it = self.maxIter - 1
        sizes = np.diff(store.offsets[: n_it + 1]).astype(np.int64)
        # per-rank iteration counts, then per-iteration segment counts of all ranks (two small exchanges)
        cnt = np.zeros(R, dtype=np.int64)
        cnt[h.rank] = n_it
        d = h.to_device(cnt)
        h.allreduce_sum_i64(d, R)
        cnt = d.to_host()
        d.free()
        meta = np.zeros(int(cnt.sum()) + 2 * R, dtype=np.int64)
        lo = int(cnt[: h.rank].sum())
        meta[lo: lo + n_it] = sizes
        meta[int(cnt.sum()) + 2 * h.rank] = store.x_dtype
        meta[int(cnt.sum()) + 2 * h.rank + 1] = store.D
        d = h.to_device(meta)
        h.allreduce_sum_i64(d, meta.size)
        meta = d.to_host()
        d.free()
        all_sizes, tail = meta[: int(cnt.sum())], meta[int(cnt.sum()):].reshape(R, 2)
        assert (tail[:, 0] == store.x_dtype).all() and (tail[:, 1] == store.D).all(), "ranks staged different frame layouts"
        rows = np.array([all_sizes[int(cnt[:r].sum()): int(cnt[: r + 1].sum())].sum() for r in range(R)], dtype=np.int64)
        n_local = int(rows[h.rank])
        esz = 4 if store.x_dtype == _abi.F32 else 8
        roots = [0]   # the frames go to rank 0
        root = h.rank in roots
        N = int(rows.sum())
        Xc = h.alloc(max(N, 1) * store.D, np.float32 if esz == 4 else np.float64) if root else None
        p0 = h.alloc(max(N, 1) * store.ndim, np.float64) if root else None
        w = h.alloc(max(N, 1), np.float64) if root else None
        for r in roots:
            h.gather_to_root(store.Xc, n_local * store.D * esz, Xc, rows * store.D * esz, r)
            h.gather_to_root(store.p0, n_local * store.ndim * 8, p0, rows * store.ndim * 8, r)
            h.gather_to_root(store.w, n_local * 8, w, rows * 8, r)
        h.sync()
        if root:
            offsets = np.concatenate([[0], np.cumsum(all_sizes)]).astype(np.int64)
            ts = DeviceStore(h, store.D, store.ndim, store.x_dtype, offsets, None, Xc, p0, None, w)
            self._train_store = (ts, len(all_sizes) + 1)
        else:
            self._train_store = (None, 0)
        return self._train_store

    def attach_store(self, store: DeviceStore, maxIter=None):
        """Use frames that are already resident (bench.py generates shards straight into HBM)."""
        self._store = store
        self._train_store = None
        if maxIter is not None:
            self.maxIter = maxIter
            self.numSegments = np.diff(store.offsets).astype(float)[:maxIter]

"""ctypes binding of the C ABI in ``include/hamsm.h``.

``load()`` returns the product library (``csrc/libhamsm_hip.so``, HIP for gfx950) and raises when
it is missing -- there is no CPU fallback.  ``Hamsm`` wraps *a* loaded library; tests construct it
around the oracle's CDLL to validate the host logic without a GPU (the oracle exports the same ABI).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

__all__ = ["Hamsm", "HamsmError", "HamsmRangeError", "HamsmUnfitError", "load", "library_path", "F32", "F64", "SYMBOLS"]

F32, F64 = 0, 1
_HERE = os.path.dirname(os.path.abspath(__file__))

_vp, _i32, _i64, _f64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_size_t
_int = C.c_int

# name -> (restype, argtypes); kept in the order of include/hamsm.h
ABI_VERSION = 10  # HAMSM_ABI_VERSION of include/hamsm.h

SYMBOLS = {
    "hamsm_abi_version": (_int, []),
    "hamsm_backend_name": (C.c_char_p, []),
    "hamsm_create": (_int, [_int, C.POINTER(_vp)]),
    "hamsm_destroy": (None, [_vp]),
    "hamsm_last_error": (C.c_char_p, [_vp]),
    "hamsm_malloc": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "hamsm_free": (_int, [_vp, _vp]),
    "hamsm_memcpy_h2d": (_int, [_vp, _vp, _vp, _sz]),
    "hamsm_memcpy_d2h": (_int, [_vp, _vp, _vp, _sz]),
    "hamsm_memset": (_int, [_vp, _vp, _int, _sz]),
    "hamsm_malloc_host": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "hamsm_free_host": (_int, [_vp, _vp]),
    "hamsm_memcpy_h2d_async": (_int, [_vp, _vp, _vp, _sz]),
    "hamsm_copy_sync": (_int, [_vp]),
    "hamsm_host_block_alloc": (_int, [_sz, C.POINTER(_vp)]),
    "hamsm_host_block_free": (_int, [_vp]),
    "hamsm_widen_f32": (_int, [_vp, _vp, _vp, _i64]),
    "hamsm_nan_rows": (_int, [_vp, _vp, _int, _i64, _int, _i64, _vp]),
    "hamsm_host_copy_parts": (_int, [_vp, _i64, _vp, _vp, _vp, _int]),
    "hamsm_host_read_parts": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _int]),
    "hamsm_sync": (_int, [_vp]),
    "hamsm_set_geometry": (_int, [_vp, _int, _int, _int]),
    "hamsm_set_mapper_rectilinear": (_int, [_vp, _vp, _vp]),
    "hamsm_set_mapper_voronoi": (_int, [_vp, _int, _vp]),
    "hamsm_set_mapper_external": (_int, [_vp]),
    "hamsm_set_frame_bins": (_int, [_vp, _vp]),
    "hamsm_set_frame_bins_host": (_int, [_vp, _vp, _i64]),
    "hamsm_set_remap": (_int, [_vp, _vp]),
    "hamsm_set_bounds": (_int, [_vp, _vp, _vp]),
    "hamsm_set_centroids": (_int, [_vp, _int, _int, _vp, _vp, _vp]),
    "hamsm_get_centroids": (_int, [_vp, _int, _vp, _vp, _vp]),
    "hamsm_num_centroids": (_int, [_vp, _int]),
    "hamsm_classify": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "hamsm_project": (_int, [_vp, _vp, _int, _i64, _i64, _int, _vp, _vp, _int, _vp, _vp]),
    "hamsm_featurize_dist": (_int, [_vp, _vp, _int, _i64, _i64, _int, _vp, _int, _vp, _vp, _i64]),
    "hamsm_pca_begin": (_int, [_vp, _int]),
    "hamsm_pca_accumulate": (_int, [_vp, _vp, _int, _i64, _i64]),
    "hamsm_pca_finish": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "hamsm_assign": (_int, [_vp, _vp, _int, _i64, _vp, _i64, _vp, _vp]),
    "hamsm_assign_pair": (_int, [_vp, _vp, _vp, _int, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "hamsm_labels_to_host": (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "hamsm_predict_host": (_int, [_vp, _vp, _int, _vp, _i64, _vp, _vp]),
    "hamsm_flux_accumulate": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp, _int, _vp, _i64]),
    "hamsm_scale": (_int, [_vp, _vp, _i64, _f64]),
    "hamsm_cluster_stats": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _vp]),
    "hamsm_group_by_label": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "hamsm_pattern_csr": (_int, [_vp, _vp, _int, _vp, _vp, _i64, _vp]),
    "hamsm_np_sum": (_int, [_vp, _vp, _i64, _vp]),
    "hamsm_permute_scale": (_int, [_vp, _vp, _int, _vp, _vp, _f64, _int]),
    "hamsm_tmatrix_csr": (_int, [_vp, _vp, _int, _vp, _int, _vp, _int, _vp, _vp, _vp, _i64, _vp]),
    "hamsm_train_prepass": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "hamsm_train_build_rows": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "hamsm_kmeans_seed": (_int, [_vp, _int, _vp, _int, _i64, _vp, _i64, _vp, _vp, _i64, _int, _i64, _vp, _int, _vp]),
    "hamsm_minibatch_step": (_int, [_vp, _int, _vp, _int, _i64, _vp, _i64, _vp, _vp]),
    "hamsm_get_counts": (_int, [_vp, _int, _vp]),
    "hamsm_reassign": (_int, [_vp, _int, _vp, _int, _i64, _vp, _vp, _vp, _int, _f64]),
    "hamsm_train_build_rows_batched": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "hamsm_kmeans_seed_multi": (_int, [_vp, _int, _vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hamsm_kmeans_seed_chosen": (_int, [_vp, _int, _vp, _i64]),
    "hamsm_minibatch_run": (_int, [_vp, _vp, _int, _i64, _vp, _vp, _vp, _i64, _f64, _vp, _i64, _vp, _vp, _vp]),
    "hamsm_comm_unique_id": (_int, [_vp]),
    "hamsm_comm_init": (_int, [_vp, _int, _int, _vp]),
    "hamsm_comm_destroy": (_int, [_vp]),
    "hamsm_allreduce_sum_f64": (_int, [_vp, _vp, _i64]),
    "hamsm_allreduce_min_f64": (_int, [_vp, _vp, _i64]),
    "hamsm_allreduce_max_f64": (_int, [_vp, _vp, _i64]),
    "hamsm_allreduce_sum_i64": (_int, [_vp, _vp, _i64]),
    "hamsm_bcast": (_int, [_vp, _vp, _sz, _int]),
    "hamsm_gather_to_root": (_int, [_vp, _vp, _sz, _vp, _vp, _int]),
    "hamsm_gather_rows": (_int, [_vp, _vp, _sz, _vp, _i64, _vp]),
    "hamsm_exchange": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "hamsm_bcast_model": (_int, [_vp, _int]),
    "hamsm_bcast_strata": (_int, [_vp, _vp]),
    "hamsm_timers": (_int, [_vp, _vp, _int]),
    "hamsm_step_kernel_ms": (_int, [_vp, _int, _vp, _vp, _int]),
    "hamsm_last_kernel_ms": (_int, [_vp, _int, _vp]),
}


class HamsmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hamsm error {code}: {msg}")
        self.code = code


class HamsmRangeError(HamsmError, ValueError):
    """HAMSM_ERANGE: a label / bin outside its range -- where the reference gets a ValueError from scipy's coo_matrix
    (`_fluxmatrix.py:143-158`; quirk Q9 makes cleaned models hit it) or from the bin mapper"""


class HamsmUnfitError(HamsmError, AssertionError):
    """HAMSM_EUNFIT: a frame maps to a stratum without centroids -- the reference's `assert hasattr(..., "cluster_centers_")`
    (`stratified_clustering.py:187-189`)"""


_ERROR_CLASSES = {-5: HamsmRangeError, -4: HamsmUnfitError}


def library_path() -> str:
    return os.path.join(_HERE, "csrc", "libhamsm_hip.so")


def load() -> C.CDLL:
    """The product library.  Fails loudly when the HIP extension has not been built."""
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C msm_we_amd/csrc).  msm_we_amd has no CPU fallback."
        )
    return C.CDLL(path)


def bind(lib: C.CDLL) -> C.CDLL:
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    return lib


def _hp(a):
    """host pointer of a numpy array (or None)"""
    if a is None:
        return None
    return a.ctypes.data_as(_vp)


class DevBuf:
    """A device allocation owned through the ABI (freed with the ctx or explicitly)."""

    __slots__ = ("h", "ptr", "nbytes", "dtype", "shape")

    def __init__(self, h, ptr, nbytes, dtype=None, shape=None):
        self.h, self.ptr, self.nbytes, self.dtype, self.shape = h, ptr, nbytes, dtype, shape

    def free(self):
        if self.ptr:
            ptr, self.ptr = self.ptr, None
            if getattr(self.h, "ctx", None):
                self.h._free(ptr)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __deepcopy__(self, memo):
        return self  # device buffers are shared by reference (never written after creation by the mixins)

    def __copy__(self):
        return self  # one owner per block: a shallow copy must not become a second one

    def __reduce__(self):
        raise TypeError("a DevBuf is a live device allocation and cannot be pickled (modelWE / StratifiedClusters pickle "
                        "their state as NumPy arrays and drop device buffers)")

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        self.h.d2h(out, self)
        return out

    def offset(self, elems):
        """device address `elems` elements (of self.dtype, row-major flat) past the start"""
        return self.ptr + int(elems) * np.dtype(self.dtype).itemsize


def _dp(b):
    if b is None:
        return None
    if isinstance(b, DevBuf):
        return b.ptr
    return int(b)


class HostBlock:
    """A page-locked host block of the PROCESS (`hamsm_host_block_alloc`, ABI 9) as the owner of NumPy views: `array()` returns an
    ndarray whose `base` chain ends in the block, so the memory lives exactly as long as some view of it does -- independent of
    any context.  Page-locking costs ~0.3 ms per MB, so released blocks are kept for the next request of the same size (a build
    discretizes the same run up to four times, block validation more often): at most `POOL_BLOCKS` per size and `POOL_BYTES` in
    all, the rest is unlocked and freed."""

    POOL_BLOCKS, POOL_BYTES = 2, 4 << 30
    _pool: dict = {}          # (library handle, nbytes) -> [addresses]
    _pooled_bytes = 0

    def __init__(self, lib, nbytes):
        self.lib, self.nbytes = lib, max(int(nbytes), 1)
        free = HostBlock._pool.get((lib._handle, self.nbytes))
        if free:
            self.ptr = free.pop()
            HostBlock._pooled_bytes -= self.nbytes
        else:
            p = _vp()
            rc = lib.hamsm_host_block_alloc(self.nbytes, C.byref(p))
            if rc != 0 or not p.value:
                raise MemoryError(f"hamsm_host_block_alloc({self.nbytes}) failed with status {rc}")
            self.ptr = p.value
        self.__array_interface__ = {"data": (self.ptr, False), "shape": (self.nbytes,), "typestr": "|u1", "version": 3}

    @classmethod
    def array(cls, lib, shape, dtype) -> np.ndarray:
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(x) for x in shape)
        count = int(np.prod(shape, dtype=np.int64))
        blk = cls(lib, count * np.dtype(dtype).itemsize)
        return np.asarray(blk)[: count * np.dtype(dtype).itemsize].view(dtype).reshape(shape)

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if not ptr:
            return
        try:
            free = HostBlock._pool.setdefault((self.lib._handle, self.nbytes), [])
            if len(free) < HostBlock.POOL_BLOCKS and HostBlock._pooled_bytes + self.nbytes <= HostBlock.POOL_BYTES:
                free.append(ptr)
                HostBlock._pooled_bytes += self.nbytes
            else:
                self.lib.hamsm_host_block_free(ptr)
        except Exception:  # noqa: BLE001  (interpreter shutdown: the process is going away with its mappings)
            pass

    @classmethod
    def drain(cls):
        """unlock and free every pooled block"""
        libs = {}
        for (hd, nb), free in list(cls._pool.items()):
            while free:
                ptr = free.pop()
                cls._pooled_bytes -= nb
                libs.setdefault(hd, C.CDLL(None, handle=hd)).hamsm_host_block_free(_vp(ptr))
        cls._pool.clear()


class Hamsm:
    """One context (one GPU).  Thin, 1:1 with the C ABI; argument meaning as in hamsm.h."""

    def __init__(self, lib: C.CDLL | None = None, device: int = 0):
        self.lib = bind(lib if lib is not None else load())
        if self.lib.hamsm_abi_version() != ABI_VERSION:
            raise ImportError("hamsm ABI version mismatch")
        ctx = _vp()
        rc = self.lib.hamsm_create(int(device), C.byref(ctx))
        if rc != 0:
            raise HamsmError(rc, "hamsm_create failed")
        self.ctx = ctx
        self.device = device
        self.D = self.ndim = self.S = None
        self.nranks, self.rank = 1, 0

    # -- plumbing ---------------------------------------------------------------------------
    @property
    def backend(self) -> str:
        return self.lib.hamsm_backend_name().decode()

    def _ck(self, rc):
        if rc != 0:
            raise _ERROR_CLASSES.get(rc, HamsmError)(rc, self.lib.hamsm_last_error(self.ctx).decode(errors="replace"))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.hamsm_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __deepcopy__(self, memo):
        return self  # a context is shared, never duplicated: two owners of one handle would destroy it twice

    __copy__ = lambda self: self  # noqa: E731

    def __reduce__(self):
        raise TypeError("a Hamsm context is a live GPU handle and cannot be pickled; pickle the model and `attach(h)` after loading")

    def alloc(self, shape, dtype) -> DevBuf:
        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = _vp()
        self._ck(self.lib.hamsm_malloc(self.ctx, max(nbytes, 1), C.byref(p)))
        return DevBuf(self, p.value, nbytes, np.dtype(dtype), shape)

    def _free(self, ptr):
        self._ck(self.lib.hamsm_free(self.ctx, ptr))

    def to_device(self, a: np.ndarray) -> DevBuf:
        a = np.ascontiguousarray(a)
        b = self.alloc(a.shape, a.dtype)
        if a.nbytes:
            self._ck(self.lib.hamsm_memcpy_h2d(self.ctx, b.ptr, _hp(a), a.nbytes))
        return b

    def h2d(self, dst, a: np.ndarray):
        a = np.ascontiguousarray(a)
        if a.nbytes:
            self._ck(self.lib.hamsm_memcpy_h2d(self.ctx, _dp(dst), _hp(a), a.nbytes))

    def d2h(self, out: np.ndarray, src, nbytes=None):
        assert out.flags.c_contiguous
        n = out.nbytes if nbytes is None else nbytes
        if n:
            self._ck(self.lib.hamsm_memcpy_d2h(self.ctx, _hp(out), _dp(src), n))
        return out

    def pinned(self, shape, dtype) -> np.ndarray:
        """a page-locked host array (freed by `free_pinned`, or with the context)"""
        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = _vp()
        self._ck(self.lib.hamsm_malloc_host(self.ctx, max(nbytes, 1), C.byref(p)))
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
        self.__dict__.setdefault("_pinned", {})[a.ctypes.data] = p.value
        return a

    def free_pinned(self, a: np.ndarray):
        ptr = self.__dict__.get("_pinned", {}).pop(a.ctypes.data, None)
        if ptr is not None and getattr(self, "ctx", None):
            self._ck(self.lib.hamsm_free_host(self.ctx, ptr))

    def h2d_async(self, dst, a: np.ndarray, nbytes=None):
        """upload from a PINNED array on the copy stream; `a` must stay untouched until `copy_sync`"""
        n = a.nbytes if nbytes is None else int(nbytes)
        if n:
            self._ck(self.lib.hamsm_memcpy_h2d_async(self.ctx, _dp(dst), _hp(a), n))

    def copy_sync(self):
        self._ck(self.lib.hamsm_copy_sync(self.ctx))

    def widen_f32(self, d_src, d_dst, count):
        """float32 frames resident in HBM -> float64 (exact), on the compute stream"""
        self._ck(self.lib.hamsm_widen_f32(self.ctx, _dp(d_src), _dp(d_dst), int(count)))

    def host_copy_parts(self, src_ptrs, dst_ptrs, nbytes, threads=1):
        """native threads copy part i (nbytes[i] bytes) from address src_ptrs[i] to dst_ptrs[i]; the interpreter lock is released
        for the duration.  The caller keeps the arrays behind the addresses alive."""
        src = np.ascontiguousarray(src_ptrs, dtype=np.uintp)
        dst = np.ascontiguousarray(dst_ptrs, dtype=np.uintp)
        nb = np.ascontiguousarray(nbytes, dtype=np.int64)
        assert len(src) == len(dst) == len(nb)
        self._ck(self.lib.hamsm_host_copy_parts(self.ctx, len(nb), _hp(src), _hp(dst), _hp(nb), int(threads)))

    def host_read_parts(self, fds, file_offsets, dst_ptrs, nbytes, threads=1):
        """native threads `pread` part i (nbytes[i] bytes at file_offsets[i] of descriptor fds[i]) to address dst_ptrs[i]"""
        fd = np.ascontiguousarray(fds, dtype=np.int32)
        fo = np.ascontiguousarray(file_offsets, dtype=np.int64)
        dst = np.ascontiguousarray(dst_ptrs, dtype=np.uintp)
        nb = np.ascontiguousarray(nbytes, dtype=np.int64)
        assert len(fd) == len(fo) == len(dst) == len(nb)
        self._ck(self.lib.hamsm_host_read_parts(self.ctx, len(nb), _hp(fd), _hp(fo), _hp(dst), _hp(nb), int(threads)))

    def nan_rows(self, d_X, x_dtype, ldx, D, n, d_flags):
        """d_flags[i] (uint8) = row i of d_X holds a NaN; on the compute stream"""
        self._ck(self.lib.hamsm_nan_rows(self.ctx, _dp(d_X), int(x_dtype), int(ldx), int(D), int(n), _dp(d_flags)))

    def zeros(self, shape, dtype) -> DevBuf:
        b = self.alloc(shape, dtype)
        self.memset(b, 0, b.nbytes)
        return b

    def memset(self, dst, byte, nbytes):
        if nbytes:
            self._ck(self.lib.hamsm_memset(self.ctx, _dp(dst), int(byte), int(nbytes)))

    def sync(self):
        self._ck(self.lib.hamsm_sync(self.ctx))

    # -- model state ------------------------------------------------------------------------
    def set_geometry(self, D, pcoord_ndim, n_strata):
        self._ck(self.lib.hamsm_set_geometry(self.ctx, int(D), int(pcoord_ndim), int(n_strata)))
        self.D, self.ndim, self.S = int(D), int(pcoord_ndim), int(n_strata)

    def set_mapper_rectilinear(self, boundaries):
        bs = [np.ascontiguousarray(b, dtype=np.float32) for b in boundaries]
        nb = np.array([len(b) for b in bs], dtype=np.int32)
        flat = np.concatenate(bs).astype(np.float32)
        self._ck(self.lib.hamsm_set_mapper_rectilinear(self.ctx, _hp(nb), _hp(flat)))

    def set_mapper_voronoi(self, centers):
        cs = np.ascontiguousarray(centers, dtype=np.float32).reshape(-1, self.ndim)
        self._ck(self.lib.hamsm_set_mapper_voronoi(self.ctx, cs.shape[0], _hp(cs)))

    def set_mapper_external(self):
        self._ck(self.lib.hamsm_set_mapper_external(self.ctx))

    def set_frame_bins(self, d_bins):
        """external mapper: the bins of the frames of the NEXT classify / assign / train_prepass call (device int32)"""
        self._ck(self.lib.hamsm_set_frame_bins(self.ctx, _dp(d_bins)))

    def set_frame_bins_host(self, bins):
        b = np.ascontiguousarray(bins, dtype=np.int32)
        self._ck(self.lib.hamsm_set_frame_bins_host(self.ctx, _hp(b), b.size))

    def set_remap(self, remap):
        r = np.ascontiguousarray(remap, dtype=np.int32)
        assert r.shape == (self.S,)
        self._ck(self.lib.hamsm_set_remap(self.ctx, _hp(r)))

    def set_bounds(self, basis, target):
        b = np.ascontiguousarray(basis, dtype=np.float64).reshape(self.ndim, 2)
        t = np.ascontiguousarray(target, dtype=np.float64).reshape(self.ndim, 2)
        self._ck(self.lib.hamsm_set_bounds(self.ctx, _hp(b), _hp(t)))

    def set_centroids(self, stratum, Cmat, csq=None, counts=None):
        if Cmat is None or len(Cmat) == 0:
            self._ck(self.lib.hamsm_set_centroids(self.ctx, int(stratum), 0, None, None, None))
            return
        Cm = np.ascontiguousarray(Cmat, dtype=np.float64)
        assert Cm.ndim == 2 and Cm.shape[1] == self.D, (Cm.shape, self.D)
        cs = None if csq is None else np.ascontiguousarray(csq, dtype=np.float64)
        ct = None if counts is None else np.ascontiguousarray(counts, dtype=np.float64)
        self._ck(self.lib.hamsm_set_centroids(self.ctx, int(stratum), Cm.shape[0], _hp(Cm), _hp(cs), _hp(ct)))

    def num_centroids(self, stratum) -> int:
        k = self.lib.hamsm_num_centroids(self.ctx, int(stratum))
        if k < 0:
            self._ck(k)
        return k

    def get_centroids(self, stratum):
        k = self.num_centroids(stratum)
        Cm = np.empty((k, self.D))
        counts = np.empty(k)
        csq = np.empty(k)
        if k:
            self._ck(self.lib.hamsm_get_centroids(self.ctx, int(stratum), _hp(Cm), _hp(counts), _hp(csq)))
        return Cm, counts, csq

    # -- ops --------------------------------------------------------------------------------
    def classify(self, d_pcoord, n, apply_remap, d_bin, d_flags):
        self._ck(self.lib.hamsm_classify(self.ctx, _dp(d_pcoord), int(n), int(apply_remap), _dp(d_bin), _dp(d_flags)))

    def project(self, d_X, x_dtype, ldx, n, W, mean=None, scale=None, d_Y=None):
        """sklearn's PCA.transform on device rows: -> DevBuf [n][d] float64"""
        W = np.ascontiguousarray(W, dtype=np.float64)
        d, Draw = W.shape
        mean = None if mean is None else np.ascontiguousarray(mean, dtype=np.float64)
        scale = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64)
        assert mean is None or mean.shape == (Draw,)
        assert scale is None or scale.shape == (d,)
        if d_Y is None:
            d_Y = self.alloc(max(int(n), 1) * d, np.float64)
        self._ck(self.lib.hamsm_project(self.ctx, _dp(d_X), int(x_dtype), int(ldx), int(n), int(Draw), _hp(mean), _hp(W), int(d),
                                        _hp(scale), _dp(d_Y)))
        return d_Y

    def pca_begin(self, Draw):
        """start a streaming PCA fit over rows of `Draw` raw features (hamsm.h f3)"""
        self._pca_D = int(Draw)
        self._ck(self.lib.hamsm_pca_begin(self.ctx, int(Draw)))

    def pca_accumulate(self, d_X, x_dtype, ldx, n):
        self._ck(self.lib.hamsm_pca_accumulate(self.ctx, _dp(d_X), int(x_dtype), int(ldx), int(n)))

    def pca_finish(self):
        """-> (n, shift [D], S1 [D], S2 [D][D]): sums of the rows seen, shifted by the first one"""
        D = self._pca_D
        n = C.c_int64(0)
        shift, S1, S2 = np.empty(D), np.empty(D), np.empty((D, D))
        self._ck(self.lib.hamsm_pca_finish(self.ctx, C.byref(n), _hp(shift), _hp(S1), _hp(S2)))
        return int(n.value), shift, S1, S2

    def featurize_dist(self, d_coords, c_dtype, ldc, n, n_atoms, idx, ref, d_X=None):
        """per-atom distance to a reference structure (the reference's test featurizer): -> DevBuf [n][len(idx)] float64"""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        ref = np.ascontiguousarray(ref, dtype=np.float32).reshape(len(idx), 3)
        if d_X is None:
            d_X = self.alloc(max(int(n), 1) * len(idx), np.float64)
        self._ck(self.lib.hamsm_featurize_dist(self.ctx, _dp(d_coords), int(c_dtype), int(ldc), int(n), int(n_atoms), _hp(idx),
                                               len(idx), _hp(ref), _dp(d_X), len(idx)))
        return d_X

    def assign(self, d_X, x_dtype, ldx, d_pcoord, n, d_labels, d_seen=None):
        self._ck(self.lib.hamsm_assign(self.ctx, _dp(d_X), int(x_dtype), int(ldx), _dp(d_pcoord), int(n),
                                       _dp(d_labels), _dp(d_seen)))

    def assign_pair(self, d_Xp, d_Xc, x_dtype, ldx, d_p0, d_p1, n, d_parent, d_child, d_seen=None):
        """both sides of the same segments: `assign(d_Xp, d_p0 -> d_parent)` then `assign(d_Xc, d_p1 -> d_child)`, enqueued
        back to back and checked once (not with an external mapper)"""
        self._ck(self.lib.hamsm_assign_pair(self.ctx, _dp(d_Xp), _dp(d_Xc), int(x_dtype), int(ldx), _dp(d_p0), _dp(d_p1), int(n),
                                            _dp(d_parent), _dp(d_child), _dp(d_seen)))

    def labels_to_host(self, d_parent, d_child, n, child=True, pairs=True):
        """-> (child int64[n] | None, pairs int64[n, 2] | None): the discretization in the reference's layout
        (`_clustering.py:1226-1240`), widened and interleaved on the device and copied into page-locked process-level blocks
        (`HostBlock`: the arrays are views of them and keep them alive -- also past this context's `close()`)."""
        n = int(n)
        a_child = HostBlock.array(self.lib, n, np.int64) if child else None
        a_pairs = HostBlock.array(self.lib, (n, 2), np.int64) if pairs else None
        self._ck(self.lib.hamsm_labels_to_host(self.ctx, _dp(d_parent), _dp(d_child), n,
                                               _hp(a_child) if child else None, _hp(a_pairs) if pairs else None))
        return a_child, a_pairs

    def predict_host(self, X: np.ndarray, pcoord: np.ndarray, want_seen=False):
        X = np.ascontiguousarray(X)
        if X.dtype == np.float32:
            dt = F32
        else:
            X = np.ascontiguousarray(X, dtype=np.float64)
            dt = F64
        p = np.ascontiguousarray(pcoord, dtype=np.float64).reshape(-1, self.ndim)
        n = X.shape[0]
        assert X.ndim == 2 and X.shape[1] == self.D and p.shape[0] == n, (X.shape, p.shape, self.D)
        labels = np.empty(n, dtype=np.int32)
        seen = np.zeros((2, self.S), dtype=np.int32) if want_seen else None
        self._ck(self.lib.hamsm_predict_host(self.ctx, _hp(X), dt, _hp(p), n, _hp(labels), _hp(seen)))
        return (labels, seen) if want_seen else labels

    def flux_accumulate(self, d_parent, d_child, d_p0, d_p1, d_w, n, K, d_F, mode=0, d_iter_offsets=None, n_iters=0):
        self._ck(self.lib.hamsm_flux_accumulate(self.ctx, _dp(d_parent), _dp(d_child), _dp(d_p0), _dp(d_p1), _dp(d_w),
                                                int(n), int(K), _dp(d_F), int(mode), _dp(d_iter_offsets), int(n_iters)))

    def scale(self, d_F, count, divisor):
        self._ck(self.lib.hamsm_scale(self.ctx, _dp(d_F), int(count), float(divisor)))

    def cluster_stats(self, d_labels, d_pcoord, n, nstates, d_sum, d_min, d_max, d_count):
        self._ck(self.lib.hamsm_cluster_stats(self.ctx, _dp(d_labels), _dp(d_pcoord), int(n), int(nstates),
                                              _dp(d_sum), _dp(d_min), _dp(d_max), _dp(d_count)))

    def group_by_label(self, d_labels, n, nstates):
        """-> (order int32[n] host, indptr int64[nstates + 1]): segment indices grouped by label, stable"""
        indptr = np.zeros(nstates + 1, dtype=np.int64)
        d_order = self.alloc(max(int(n), 1), np.int32)
        try:
            self._ck(self.lib.hamsm_group_by_label(self.ctx, _dp(d_labels), int(n), int(nstates), _dp(d_order), _hp(indptr)))
            order = np.empty(int(n), dtype=np.int32)
            self.d2h(order, d_order, order.nbytes)
        finally:
            d_order.free()
        return order, indptr

    def tmatrix_csr(self, d_F, nstates, basis_idx, target_idx):
        b = np.ascontiguousarray(basis_idx, dtype=np.int32)
        t = np.ascontiguousarray(target_idx, dtype=np.int32)
        indptr = np.empty(nstates + 1, dtype=np.int64)
        nnz = _i64(0)
        # count pass, then arrays of exactly that size (not nstates^2: 300 MB of host buffers at 5002 states)
        self._ck(self.lib.hamsm_tmatrix_csr(self.ctx, _dp(d_F), int(nstates), _hp(b), len(b), _hp(t), len(t), _hp(indptr), None,
                                            None, 0, C.byref(nnz)))
        k = nnz.value
        indices = np.empty(max(k, 1), dtype=np.int32)
        data = np.empty(max(k, 1), dtype=np.float64)
        self._ck(self.lib.hamsm_tmatrix_csr(self.ctx, _dp(d_F), int(nstates), _hp(b), len(b), _hp(t), len(t),
                                            _hp(indptr), _hp(indices), _hp(data), max(k, 1), C.byref(nnz)))
        return indptr, indices[:k], data[:k]

    def pattern_csr(self, d_F, nstates):
        """sparsity pattern of a dense device matrix: (indptr int64[n+1], indices int32[nnz])"""
        indptr = np.empty(nstates + 1, dtype=np.int64)
        nnz = _i64(0)
        self._ck(self.lib.hamsm_pattern_csr(self.ctx, _dp(d_F), int(nstates), _hp(indptr), None, 0, C.byref(nnz)))
        k = nnz.value
        indices = np.empty(max(k, 1), dtype=np.int32)
        if k:
            self._ck(self.lib.hamsm_pattern_csr(self.ctx, _dp(d_F), int(nstates), _hp(indptr), _hp(indices), k, C.byref(nnz)))
        return indptr, indices[:k]

    def np_sum(self, d_A, count) -> float:
        v = _f64(0.0)
        self._ck(self.lib.hamsm_np_sum(self.ctx, _dp(d_A), int(count), C.byref(v)))
        return v.value

    def permute_scale(self, d_F, nstates, perm, d_out, divisor=1.0, transpose=False):
        pm = np.ascontiguousarray(perm, dtype=np.int32)
        assert pm.shape == (nstates,)
        self._ck(self.lib.hamsm_permute_scale(self.ctx, _dp(d_F), int(nstates), _hp(pm), _dp(d_out), float(divisor), int(transpose)))

    def train_prepass(self, d_pcoord, n, iter_offsets, d_bin):
        offs = np.ascontiguousarray(iter_offsets, dtype=np.int64)
        n_iters = len(offs) - 1
        hist = np.zeros((n_iters, self.S), dtype=np.int64)
        self._ck(self.lib.hamsm_train_prepass(self.ctx, _dp(d_pcoord), int(n), _hp(offs), n_iters, _dp(d_bin), _hp(hist)))
        return hist

    def train_build_rows(self, d_bin, n, call_remap, d_rows):
        cr = np.ascontiguousarray(call_remap, dtype=np.int32)
        counts = np.zeros(self.S, dtype=np.int64)
        starts = np.zeros(self.S, dtype=np.int64)
        self._ck(self.lib.hamsm_train_build_rows(self.ctx, _dp(d_bin), int(n), _hp(cr), _dp(d_rows), _hp(counts), _hp(starts)))
        return counts, starts

    def kmeans_seed(self, stratum, d_X, x_dtype, ldx, d_rows, n_b, d_w, sub, K_s, first, u, L, want_chosen=False):
        subh = None if sub is None else np.ascontiguousarray(sub, dtype=np.int64)
        uh = np.ascontiguousarray(u, dtype=np.float64)
        assert uh.size == (K_s - 1) * L
        chosen = np.empty(K_s, dtype=np.int64) if want_chosen else None
        self._ck(self.lib.hamsm_kmeans_seed(self.ctx, int(stratum), _dp(d_X), int(x_dtype), int(ldx), _dp(d_rows), int(n_b),
                                            _dp(d_w), _hp(subh), 0 if subh is None else len(subh), int(K_s), int(first),
                                            _hp(uh), int(L), _hp(chosen)))
        return chosen

    def minibatch_step(self, stratum, d_X, x_dtype, ldx, d_rows, n_b, d_w=None, d_labels=None):
        self._ck(self.lib.hamsm_minibatch_step(self.ctx, int(stratum), _dp(d_X), int(x_dtype), int(ldx), _dp(d_rows),
                                               int(n_b), _dp(d_w), _dp(d_labels)))

    def get_counts(self, stratum):
        k = self.num_centroids(stratum)
        out = np.empty(k)
        if k:
            self._ck(self.lib.hamsm_get_counts(self.ctx, int(stratum), _hp(out)))
        return out

    def reassign(self, stratum, d_X, x_dtype, ldx, d_rows, clusters, batch_rows, new_count):
        cl = np.ascontiguousarray(clusters, dtype=np.int32)
        br = None if batch_rows is None else np.ascontiguousarray(batch_rows, dtype=np.int64)
        self._ck(self.lib.hamsm_reassign(self.ctx, int(stratum), _dp(d_X), int(x_dtype), int(ldx), _dp(d_rows), _hp(cl),
                                         _hp(br), len(cl), float(new_count)))

    # -- batched training plan ----------------------------------------------------------------
    STEP_DTYPE = np.dtype([("stratum", "<i4"), ("flags", "<i4"), ("x_off", "<i8"), ("rows_off", "<i8"), ("n_b", "<i8"),
                           ("perm_off", "<i8"), ("n_perm", "<i4"), ("n_span", "<i4")])
    STOP_NONE, STOP_LOW, STOP_UNUSED, STOP_CAP, STOP_ZERO = 0, 1, 2, 3, 4

    def train_build_rows_batched(self, d_bin, group_begin, group_end, call_remap, d_rows, readback=True):
        goff = np.ascontiguousarray(np.stack([group_begin, group_end], axis=1), dtype=np.int64)
        G = len(goff)
        cr = np.ascontiguousarray(np.asarray(call_remap), dtype=np.int32).reshape(G, self.S)
        if not readback:   # the caller knows counts and starts (ABI 9): enqueue only
            self._ck(self.lib.hamsm_train_build_rows_batched(self.ctx, _dp(d_bin), G, _hp(goff), _hp(cr), _dp(d_rows), None, None))
            return None, None
        counts = np.zeros((G, self.S), dtype=np.int64)
        starts = np.zeros((G, self.S), dtype=np.int64)
        self._ck(self.lib.hamsm_train_build_rows_batched(self.ctx, _dp(d_bin), G, _hp(goff), _hp(cr), _dp(d_rows),
                                                         _hp(counts), _hp(starts)))
        return counts, starts

    def kmeans_seed_multi(self, strata, d_X, x_dtype, ldx, x_off, d_rows, rows_off, n_b, d_w, subs, K, first, us, L):
        n = len(strata)
        a32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)  # noqa: E731
        a64 = lambda v: np.ascontiguousarray(v, dtype=np.int64)  # noqa: E731
        n_sub = a64([0 if sb is None else len(sb) for sb in subs])
        sub = a64(np.concatenate([np.asarray(sb, dtype=np.int64) for sb in subs if sb is not None])
                  if any(sb is not None for sb in subs) else np.zeros(1, np.int64))
        u = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float64).ravel() for x in us] + [np.zeros(1)]))
        st, xo, ro, nb, Ks, fs, Ls = a32(strata), a64(x_off), a64(rows_off), a64(n_b), a32(K), a64(first), a32(L)
        self._ck(self.lib.hamsm_kmeans_seed_multi(self.ctx, n, _hp(st), _dp(d_X), int(x_dtype), int(ldx), _hp(xo), _dp(d_rows),
                                                  _hp(ro), _hp(nb), _dp(d_w), _hp(sub), _hp(n_sub), _hp(Ks), _hp(fs), _hp(u),
                                                  _hp(Ls)))

    def kmeans_seed_chosen(self, item, K):
        """batch indices item `item` of the last `kmeans_seed_multi` chose (parity tests; synchronises)"""
        out = np.empty(int(K), dtype=np.int64)
        self._ck(self.lib.hamsm_kmeans_seed_chosen(self.ctx, int(item), _hp(out), len(out)))
        return out

    def minibatch_run(self, d_X, x_dtype, ldx, d_rows, d_w, steps, ratio, perm=None):
        """-> (report[n_steps, 2], resume[S], reason[S]); steps without a permutation carry perm_off = -1"""
        steps = np.ascontiguousarray(steps, dtype=self.STEP_DTYPE)
        report = np.zeros((len(steps), 2), dtype=np.int32)
        resume = np.full(self.S, len(steps), dtype=np.int64)
        reason = np.zeros(self.S, dtype=np.int32)
        perm = np.zeros(0, dtype=np.int32) if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        if len(steps):
            self._ck(self.lib.hamsm_minibatch_run(self.ctx, _dp(d_X), int(x_dtype), int(ldx), _dp(d_rows), _dp(d_w),
                                                  _hp(steps), len(steps), float(ratio), _hp(perm) if len(perm) else None,
                                                  len(perm), _hp(report), _hp(resume), _hp(reason)))
        return report, resume, reason

    # -- exchange step ----------------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._ck(self.lib.hamsm_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks, rank, uid: bytes):
        assert len(uid) == 128
        self._ck(self.lib.hamsm_comm_init(self.ctx, int(nranks), int(rank), C.c_char_p(uid)))
        self.nranks, self.rank = nranks, rank

    def comm_destroy(self):
        self._ck(self.lib.hamsm_comm_destroy(self.ctx))

    def allreduce_sum_f64(self, d_buf, count):
        self._ck(self.lib.hamsm_allreduce_sum_f64(self.ctx, _dp(d_buf), int(count)))

    def allreduce_min_f64(self, d_buf, count):
        self._ck(self.lib.hamsm_allreduce_min_f64(self.ctx, _dp(d_buf), int(count)))

    def allreduce_max_f64(self, d_buf, count):
        self._ck(self.lib.hamsm_allreduce_max_f64(self.ctx, _dp(d_buf), int(count)))

    def allreduce_sum_i64(self, d_buf, count):
        self._ck(self.lib.hamsm_allreduce_sum_i64(self.ctx, _dp(d_buf), int(count)))

    def bcast(self, d_buf, nbytes, root=0):
        self._ck(self.lib.hamsm_bcast(self.ctx, _dp(d_buf), int(nbytes), int(root)))

    def gather_to_root(self, d_send, send_bytes, d_recv, recv_bytes, root=0):
        rb = np.ascontiguousarray(recv_bytes, dtype=np.int64)
        self._ck(self.lib.hamsm_gather_to_root(self.ctx, _dp(d_send), int(send_bytes), _dp(d_recv), _hp(rb), int(root)))

    def gather_rows(self, d_src, row_bytes, d_idx, n, d_dst):
        """d_dst row i = d_src row d_idx[i] (int64 indices on the device; rows of `row_bytes` = a multiple of 4 bytes)"""
        self._ck(self.lib.hamsm_gather_rows(self.ctx, _dp(d_src), int(row_bytes), _dp(d_idx), int(n), _dp(d_dst)))

    def exchange(self, d_send, send_bytes, d_recv, recv_bytes):
        """send_bytes[q] bytes to rank q (consecutive in d_send, rank order), recv_bytes[q] from rank q (consecutive in d_recv)"""
        sb = np.ascontiguousarray(send_bytes, dtype=np.int64)
        rb = np.ascontiguousarray(recv_bytes, dtype=np.int64)
        assert len(sb) == len(rb) == max(1, getattr(self, "nranks", 1))
        self._ck(self.lib.hamsm_exchange(self.ctx, _dp(d_send) if d_send is not None else None, _hp(sb),
                                         _dp(d_recv) if d_recv is not None else None, _hp(rb)))

    def bcast_strata(self, roots):
        """stratum s's model from rank roots[s] to every rank (remap table from rank 0)"""
        r = np.ascontiguousarray(roots, dtype=np.int32)
        assert r.size == self.S
        self._ck(self.lib.hamsm_bcast_strata(self.ctx, _hp(r)))

    def bcast_model(self, root=0):
        self._ck(self.lib.hamsm_bcast_model(self.ctx, int(root)))

    def timers(self, reset=False):
        out = np.zeros(8)
        self._ck(self.lib.hamsm_timers(self.ctx, _hp(out), int(reset)))
        return dict(zip(("assign", "flux", "stats", "seed", "step", "reassign", "prepass", "other"), out))

    def step_kernel_ms(self, which, reset=False):
        """(mean ms, samples) of the E-step (0) / M-step (1) training kernel over the sampled launches"""
        v, n = _f64(0.0), _i64(0)
        self._ck(self.lib.hamsm_step_kernel_ms(self.ctx, int(which), C.byref(v), C.byref(n), int(reset)))
        return v.value, n.value

    def last_kernel_ms(self, stage) -> float:
        v = _f64(0.0)
        self._ck(self.lib.hamsm_last_kernel_ms(self.ctx, int(stage), C.byref(v)))
        return v.value

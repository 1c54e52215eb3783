"""``FluxMatrixMixin``: segment-pair flux-matrix accumulation on the device.

Mirrors ``msm_we/_hamsm/_fluxmatrix.py:21-366``.  The reference builds one sparse (K+2)^2 matrix
per iteration in a Ray task and sums them densely on the driver (``:275-335``); here one kernel
scatters every segment's weight into a device-resident (K+2)^2 f64 matrix.
"""
from __future__ import annotations

import logging

import numpy as np

log = logging.getLogger("msm_we_amd")

__all__ = ["FluxMatrixMixin"]


def _lazy_matrix(name):
    """`model.<name>`: a dense (n, n) float64 host array as in the reference -- kept on the device (`_d_<name>`) by the steps
    that produce it and copied to the host the first time somebody reads it.  Assigning a host array drops the device copy."""
    host, dev = "_h_" + name, "_d_" + name

    def get(self):
        v = self.__dict__.get(host)
        if v is None and self.__dict__.get(dev) is not None:
            d = self.__dict__[dev]
            v = np.empty(d.shape, dtype=np.float64)
            self.h.d2h(v, d)
            self.__dict__[host] = v
        return v

    def set_(self, value):
        self.__dict__[host] = value
        self.__dict__[dev] = None

    return property(get, set_)


class FluxMatrixMixin:
    fluxMatrixRaw = _lazy_matrix("fluxMatrixRaw")
    fluxMatrix = _lazy_matrix("fluxMatrix")
    flux_mode = 1  # 1 = deterministic (bit-identical to the per-iteration dense sums); 0 = f64 atomics

    def _matrix_device(self, name):
        """the device copy of `fluxMatrixRaw` / `fluxMatrix` (uploaded when only a host array exists, e.g. after unpickling)"""
        d = self.__dict__.get("_d_" + name)
        if d is None:
            v = self.__dict__.get("_h_" + name)
            if v is None:
                raise RuntimeError(f"{name} has not been computed")
            d = self.h.to_device(np.ascontiguousarray(v, dtype=np.float64))
            self.__dict__["_d_" + name] = d
        return d

    def _set_matrix_device(self, name, d_buf):
        self.__dict__["_d_" + name] = d_buf
        self.__dict__["_h_" + name] = None

    def get_fluxMatrix(self, n_lag, first_iter=1, last_iter=None, iters_to_use=None, use_ray=False,
                       result_batch_size=5, progress_bar=None):
        """`_fluxmatrix.py:166-345`.  Post-condition: ``self.fluxMatrixRaw`` dense float64 (K+2, K+2)."""
        self._fluxMatrixParams = [n_lag, first_iter, last_iter, iters_to_use]
        if iters_to_use is None:
            if last_iter is None:
                last_iter = self.maxIter
            iters_to_use = range(first_iter + 1, last_iter)  # :215 (quirk Q6)
        self.n_lag = n_lag
        self.errorWeight = 0.0
        self.errorCount = 0
        # (a contiguous range stays a range: a thousand int() calls and run-merging steps cost ~0.1 ms between two launches)
        iters = iters_to_use if isinstance(iters_to_use, range) and iters_to_use.step == 1 else [int(i) for i in iters_to_use]
        h = self.h
        store = self._store
        K = self.n_clusters
        ns = K + 2
        d_F = self._flux_device(iters)
        nI = len(iters)
        if getattr(h, "nranks", 1) > 1:
            # exchange step 2: sum of the ranks' partial flux matrices (replaces the driver-side
            # `fluxMatrix + _fmatrix.todense().A` over Ray results, :325-327) and of the iteration counts
            h.allreduce_sum_f64(d_F, ns * ns)
            d_n = h.to_device(np.array([nI], dtype=np.int64))
            h.allreduce_sum_i64(d_n, 1)
            nI = int(d_n.to_host()[0])
            d_n.free()
        h.scale(d_F, ns * ns, float(nI))  # :342
        self._set_matrix_device("fluxMatrixRaw", d_F)   # stays in HBM; `self.fluxMatrixRaw` copies it out on first read

    def _flux_device(self, iters):
        """Sum of the per-iteration flux matrices of `iters` as a device (K+2)^2 buffer (caller frees)."""
        h = self.h
        store = self.stage_iterations()
        d_parent, d_child = self._labels_device()
        K = self.n_clusters
        ns = K + 2
        d_F = h.zeros((ns, ns), np.float64)
        # contiguous runs of iterations become single launches
        runs = []
        if isinstance(iters, range) and iters.step == 1:
            if len(iters):
                for it in (iters[0], iters[-1]):
                    if it < 1 or it > self.maxIter - 1:
                        raise IndexError(f"iteration {it} has no discretized pairs")  # pair_dtrajs[it-1] (:288)
                runs.append([iters[0], iters[-1]])
            iters = ()
        for it in iters:
            if it < 1 or it > self.maxIter - 1:
                raise IndexError(f"iteration {it} has no discretized pairs")  # pair_dtrajs[it-1] (:288)
            if runs and it == runs[-1][1] + 1:
                runs[-1][1] = it
            else:
                runs.append([it, it])
        for a, b in runs:
            f0, f1 = store.span(a, b)
            # array bases are unshifted: d_iter_offsets holds absolute segment positions (hamsm.h)
            h.flux_accumulate(d_parent, d_child, store.p0, store.p1, store.w_flux, f1 - f0, K, d_F,
                              self.flux_mode, store.d_offsets.offset(a - 1), b - a + 1)
        return d_F

    def organize_fluxMatrix(self, use_ray=False, progress_bar=None, **args):
        """`_fluxmatrix.py:347-366` (stratified branch; `organize_aggregated` is deprecated there, :452)."""
        if getattr(self, "clustering_method", None) != "stratified":
            raise NotImplementedError("only stratified clustering is supported")
        self.organize_stratified(use_ray, progress_bar)

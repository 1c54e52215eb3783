"""Stand-ins for the two WESTPA bin mappers the reference supports (``SUPPORTED_MAPPERS``, reference
``msm_we/_hamsm/_clustering.py:598-609``, ``find_nearest_bin :1357-1396``) for hosts where ``westpa`` is not installed, and the
marker that lets a Voronoi mapper run on the device.

The library computes bins itself for ``RectilinearBinMapper`` (float32 half-open digitize) and for a ``VoronoiBinMapper`` whose
``dfunc`` is :func:`euclidean_dfunc` (or any function carrying ``hamsm_kind = "euclidean"``).  A Voronoi mapper with any other
distance function keeps its own ``assign`` -- it runs on the host, once per staged pcoord array, and the library takes the bins
from the resulting array (``hamsm_set_mapper_external``).  Real ``westpa.core.binning`` objects work the same way: dispatch is by
class name and attributes, as in the reference.
"""
from __future__ import annotations

import numpy as np

__all__ = ["RectilinearBinMapper", "VoronoiBinMapper", "euclidean_dfunc"]


def euclidean_dfunc(coordvec, centers, *args, **kwargs):
    """westpa dfunc contract: distances from ONE coordinate vector to every centre"""
    d = np.asarray(centers, dtype=np.float64) - np.asarray(coordvec, dtype=np.float64)
    return np.sqrt(np.sum(d * d, axis=1))


euclidean_dfunc.hamsm_kind = "euclidean"


class RectilinearBinMapper:
    def __init__(self, boundaries):
        self._boundaries = [np.asarray(b, dtype=np.float32) for b in boundaries]
        self.boundaries = self._boundaries
        self.ndim = len(self._boundaries)
        self.nbins = int(np.prod([len(b) - 1 for b in self._boundaries]))

    def assign(self, coords, mask=None, output=None):
        coords = np.asarray(coords, dtype=np.float32).reshape(-1, self.ndim)
        idx = np.zeros(coords.shape[0], dtype=np.int64)
        for d, b in enumerate(self._boundaries):
            j = np.searchsorted(b, coords[:, d], side="right") - 1
            if np.any(j < 0) or np.any(j >= len(b) - 1):
                raise ValueError("coordinate outside bin space")
            idx = idx * (len(b) - 1) + j
        return idx.astype(np.uint16)


class VoronoiBinMapper:
    """nearest centre under ``dfunc(coordvec, centers, *dfargs, **dfkwargs)`` (first minimum), float32 coordinates"""

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

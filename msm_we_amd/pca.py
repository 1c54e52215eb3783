"""Exact PCA from covariance sums accumulated on the device (opt-in: ``modelWE.pca_fit = "device"``).

The reference fits scikit-learn's ``IncrementalPCA`` one WE iteration at a time (`msm_we/_hamsm/_dimensionality.py:69-108`,
driven by `:139-241`): a forked process and an SVD of ``(components + batch) x features`` per iteration, twice over the data
(a "rough" fit on a subset to find how many components explain ``variance_cutoff``, then the fit proper).  Here the same
iterations stream through ``hamsm_pca_accumulate`` (f64 matrix unit, one pass), and this module turns the sums into the
attributes of a fitted scikit-learn PCA -- the D x D symmetric eigen-problem is host work, like every solve of the package.

Not the default: ``IncrementalPCA`` with a component cap truncates at every batch, so its components are not a function of
the covariance and the two fits agree only approximately (subspace angles, not bits); the default ``pca_fit = "host"`` keeps
the reference's own estimator and its bit-identical golden.
"""
from __future__ import annotations

import numpy as np

__all__ = ["CovariancePCA"]


class CovariancePCA:
    """The attribute surface of a fitted ``sklearn.decomposition.PCA`` (what `reduceCoordinates`, pickling consumers and
    `hamsm_project` read): ``components_ [d, D]``, ``mean_``, ``explained_variance_``, ``explained_variance_ratio_``,
    ``singular_values_``, ``n_components_``, ``n_samples_seen_``, ``whiten``; ``transform`` as `_BasePCA.transform`."""

    whiten = False

    def __init__(self, n_components=None):
        self.n_components = n_components

    def fit_from_sums(self, n, shift, S1, S2):
        """n rows x; shift [D]; S1 = sum (x - shift); S2 = sum (x - shift)(x - shift)^T  (hamsm_pca_finish)"""
        n = int(n)
        if n < 2:
            raise ValueError(f"PCA needs at least two rows, got {n}")
        S1 = np.asarray(S1, dtype=np.float64)
        S2 = np.asarray(S2, dtype=np.float64)
        D = S1.shape[0]
        delta = S1 / n
        self.mean_ = np.asarray(shift, dtype=np.float64) + delta
        cov = (S2 - np.outer(S1, delta)) / (n - 1)
        cov = 0.5 * (cov + cov.T)
        ev, vec = np.linalg.eigh(cov)          # ascending
        ev, vec = ev[::-1], vec[:, ::-1]
        ev = np.clip(ev, 0.0, None)
        W = vec.T.copy()
        # sign convention of sklearn's svd_flip(u_based_decision=False): the largest-magnitude entry of a component is positive
        idx = np.argmax(np.abs(W), axis=1)
        W *= np.sign(W[np.arange(D), idx])[:, None]
        total = ev.sum()
        d = D if self.n_components is None else int(self.n_components)
        if not 1 <= d <= min(D, n):
            raise ValueError(f"n_components={d} must be between 1 and min(n_samples, n_features)={min(D, n)}")
        self.n_features_in_ = D
        self.n_samples_seen_ = n
        self.n_components_ = d
        self.components_ = W[:d]
        self.explained_variance_ = ev[:d]
        self.explained_variance_ratio_ = ev[:d] / total if total > 0 else np.zeros(d)
        self.singular_values_ = np.sqrt(ev[:d] * (n - 1))
        self.noise_variance_ = float(ev[d:].mean()) if d < min(D, n) else 0.0
        self.var_ = np.diag(cov).copy()
        return self

    def transform(self, X):
        """`sklearn/decomposition/_base.py:147-156`: X @ components_.T - mean_ @ components_.T"""
        X = np.asarray(X, dtype=np.float64)
        return X @ self.components_.T - self.mean_ @ self.components_.T

    def inverse_transform(self, Y):
        return np.asarray(Y, dtype=np.float64) @ self.components_ + self.mean_

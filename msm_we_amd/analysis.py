"""``AnalysisMixin``: transition matrix, steady state, target flux, committor.

The solve stays on the host (BASELINE.json north star) and is fed by the
GPU-built CSR transition matrix (``hamsm_tmatrix_csr``).  Semantics follow the reference's
``msm_we/_hamsm/_analysis.py:23-79, 97-282, 317-384, 527-606`` and ``utils.py:87-161``: dominant left
eigenvector refined by inverse iteration, convergence on the target flux.  Where the reference forms
an explicit sparse inverse (``utils.py:138``) this factorises once and solves (same SuperLU
factorisation, no inverse); above ``dense_eig_max`` states the dense ``eig`` seed (``:222``) is
replaced by ARPACK on the CSR matrix, and matrices denser
than 10 % are factorised with LAPACK instead of SuperLU.
"""
from __future__ import annotations

import contextlib
import logging
import os
import warnings

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.csgraph as csgraph
import scipy.sparse.linalg as spla

log = logging.getLogger("msm_we_amd")

__all__ = ["AnalysisMixin", "inverse_iteration", "is_connected"]

DENSE_LU_DENSITY = 0.10
DENSE_LU_MAX_N = 6144      # states up to which a dense LAPACK factorisation (<= 300 MB, < 1 s) is preferred over a risky sparse one


def _prefer_dense_lu(matrix, n):
    """LAPACK getrf or SuperLU for (T^T - mu I)?  Dense when the matrix is dense anyway; for the sizes this package meets
    (K + 2 <= ~5000) also whenever the pattern is not essentially banded: sparse LU of a cleaned haMSM with a percent of
    long-range entries fills in catastrophically (5002 states, 1.5 % dense: SuperLU 11.7 s, getrf 0.5 s; tools/solvebench.py),
    while a banded pattern factorises in milliseconds."""
    if matrix.nnz > DENSE_LU_DENSITY * n * n:
        return True
    if n > DENSE_LU_MAX_N:
        return False
    coo = matrix.tocoo()
    far = np.count_nonzero(np.abs(coo.row.astype(np.int64) - coo.col) > max(8, n // 16))
    return far > 0.005 * max(1, matrix.nnz)
BLAS_THREADS_MAX = 16


def _cpu_share():
    """CPUs this process may actually use: affinity mask, capped by the cgroup quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                fields = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = fields[0], float(fields[1])
            else:
                quota = fields[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = float(f.read())
            if quota not in ("max", "-1") and period > 0:
                n = min(n, max(1, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _bounded_blas(n=None):
    """OpenBLAS sizes its pool by the machine's core count (64 threads on a 256-core host): on a 16-CPU share that
    makes a 2600-state getrf 10x slower (0.49 s vs 0.045 s measured on the MI355X box; tools/lu_threads.py)."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        return contextlib.nullcontext()
    if n is not None and n <= 512:
        return threadpool_limits(limits=1, user_api="blas")  # small matrices: waking a thread pool costs more than it saves
    return threadpool_limits(limits=max(1, min(BLAS_THREADS_MAX, _cpu_share())), user_api="blas")


def is_connected(matrix, source_states, target_states, directed=True):
    """reference `utils.py:87-113`"""
    # the reference asks for all shortest-path lengths and tests them for inf; reachability is the same
    # predicate and one breadth-first sweep per source instead of Dijkstra over every edge weight
    matrix = sp.csr_matrix(matrix)
    targets = np.atleast_1d(np.asarray(target_states, dtype=np.int64))
    for s in np.atleast_1d(np.asarray(source_states, dtype=np.int64)):
        seen = np.zeros(matrix.shape[0], dtype=bool)
        seen[csgraph.breadth_first_order(matrix, int(s), directed=directed, return_predecessors=False)] = True
        if not seen[targets].all():
            return False
    return True


class _ShiftedSolver:
    """(T^T - mu I) factorised ONCE: the reference's `inverse_iteration` forms the inverse anew at every step of
    `get_steady_state`'s loop (`utils.py:138`, `_analysis.py:144-146`) although the matrix does not change -- at 2600 states
    the factorisation is 95 % of a step.  LAPACK getrf or SuperLU by `_prefer_dense_lu`; `solve` returns None when the shifted
    matrix is exactly singular (the caller then retries with mu = 0.999, as the reference does)."""

    def __init__(self, matrix, mu=1):
        self.n = n = matrix.shape[0]
        self.mu = mu
        self.lu = self.splu = None
        matrix = sp.csr_matrix(matrix)
        if _prefer_dense_lu(matrix, n):
            # cleaned haMSM matrices are often 30-50 % dense: LAPACK getrf beats SuperLU there by 2-4x
            with warnings.catch_warnings(), _bounded_blas(n):
                warnings.simplefilter("ignore", sla.LinAlgWarning)
                A = matrix.toarray().T
                A[np.diag_indices(n)] -= mu
                lu = sla.lu_factor(A, overwrite_a=True, check_finite=False)
                if np.all(np.diagonal(lu[0]) != 0):
                    self.lu = lu
        else:
            try:
                self.splu = spla.splu((matrix.T - mu * sp.eye(n)).tocsc())
            except RuntimeError:
                self.splu = None

    def solve(self, guess):
        guess = np.asarray(guess, dtype=np.float64)
        x = None
        if self.lu is not None:
            with _bounded_blas(self.n):
                x = sla.lu_solve(self.lu, guess, check_finite=False)
        elif self.splu is not None:
            x = self.splu.solve(guess)
        if x is None or not np.all(np.isfinite(x)):
            return None
        return x / sum(x)


def inverse_iteration(guess, matrix, mu=1, _solver=None):
    """One step of inverse iteration towards the eigenvalue-1 left eigenvector (reference
    `utils.py:116-161`): solve (T^T - mu I) x = guess, normalise; fall back to mu=0.999 when the
    shifted matrix is exactly singular.  `_solver`: a factorisation of the same shifted matrix to re-use."""
    solver = _solver if (_solver is not None and _solver.mu == mu) else _ShiftedSolver(matrix, mu)
    x = solver.solve(guess)
    if x is None:
        if mu != 1:
            raise RuntimeError("inverse iteration failed: shifted matrix is singular")
        log.error("When solving steady-state, failed to perform inverse iteration! Trying again with mu=0.999.")
        return inverse_iteration(guess, matrix, mu=0.999)
    return x


class AnalysisMixin:
    Tmatrix_csr = None
    pSS = None
    lagtime = None
    JtargetSS = None
    dense_eig_max = 128  # states; above this the eigenvector seed comes from ARPACK on the CSR matrix (300 states: 4 ms vs 190 ms)

    @property
    def Tmatrix(self):
        """dense (n, n) transition matrix as in the reference (`_analysis.py:79`); built from the CSR matrix the device
        produced the first time it is read"""
        d = self.__dict__.get("_Tmatrix_dense")
        if d is None and self.Tmatrix_csr is not None:
            d = self.Tmatrix_csr.toarray()
            self.__dict__["_Tmatrix_dense"] = d
        return d

    @Tmatrix.setter
    def Tmatrix(self, value):
        self.__dict__["_Tmatrix_dense"] = value
        self.Tmatrix_csr = None

    def get_Tmatrix(self):
        """`_analysis.py:23-79` on the device: row-normalise, zero rows -> self loop, target rows -> uniform over the basis,
        straight from the organised flux matrix in HBM; the result is CSR (`Tmatrix_csr`, sized by a count pass) and the
        dense `Tmatrix` of the reference is derived from it on first access."""
        h = self.h
        d_F = self._matrix_device("fluxMatrix")
        n = int(d_F.shape[0])
        indptr, indices, data = h.tmatrix_csr(d_F, n, self.indBasis, self.indTargets)
        self.Tmatrix_csr = sp.csr_matrix((data, indices, indptr), shape=(n, n))
        self.__dict__["_Tmatrix_dense"] = None

    def _T_csr(self):
        T = self.Tmatrix_csr
        if T is None:
            T = sp.csr_matrix(self.Tmatrix)
        return T

    def get_steady_state_algebraic(self, max_iters=1000, check_negative=True, set=True):
        """`_analysis.py:193-282`"""
        n = self._T_csr().shape[0]
        pSS = None
        if n > self.dense_eig_max:
            # dense eig is O(n^3) with a large constant (1 s at 670 states, 5 s at 2600); ARPACK on the device-built CSR
            # needs a handful of mat-vecs.  Deterministic start vector; dense fallback when it does not converge.
            try:
                vals, vecs = spla.eigs(self._T_csr().T.tocsc(), k=1, which="LR", v0=np.full(n, 1.0 / n), tol=1e-13,
                                       maxiter=max(20 * n, 2000))
                pSS = np.real(vecs[:, 0]).squeeze()
                if np.isclose(np.sum(pSS), 0):
                    pSS = None
            except (spla.ArpackNoConvergence, spla.ArpackError) as e:
                log.warning(f"ARPACK did not converge ({e}); falling back to dense eig")
        if pSS is None:
            with _bounded_blas(n):
                vals, vecs = np.linalg.eig(np.transpose(self.Tmatrix))
            pSS = np.real(vecs[:, np.argmax(np.real(vals))]).squeeze()
        assert not np.isclose(np.sum(pSS), 0), "Steady-state distribution sums to 0!"
        pSS = pSS / np.sum(pSS)
        if sum(pSS < 0) > 0 and max_iters > 0:
            # matrix-power correction, :236-260
            last, Tp = pSS, self.Tmatrix.copy()
            new = pSS
            for N in range(max_iters):
                new = Tp.T @ last
                if sum(new < 0) == 0:
                    break
                last = new
                Tp = np.matmul(self.Tmatrix, Tp)
            if N != max_iters - 1:
                pSS = new
        if not np.all(pSS >= 0):
            if check_negative:
                assert np.all(pSS >= 0), f"Some negative elements in steady-state distribution: {pSS}"
            else:
                log.warning("Some negative  elements in pSS... Ignoring, and setting model.pSS anyways.")
        if set:
            self.pSS = pSS
        else:
            return pSS

    def get_steady_state(self, flux_fractional_convergence=1e-4, max_iters=10):
        """`_analysis.py:97-191`"""
        T = self._T_csr()
        n = T.shape[0]
        solver = _ShiftedSolver(T, 1)   # the loop's matrix never changes: factorised once, not once per step
        # The seed.  The reference takes it from a dense `eig` (`get_steady_state_algebraic`); above `dense_eig_max` states this
        # package used ARPACK for it (0.1 s at 2661 states, more than the factorisation).  But (T^T - I) is singular by
        # construction, so ONE solve with its factors from any positive vector already is the eigenvector to rounding (3e-15 of
        # ARPACK's at 2661 states): the seed costs a pair of triangular solves, and the loop below confirms it as before.
        last_pSS = None
        if n > self.dense_eig_max:
            last_pSS = solver.solve(np.full(n, 1.0 / n))
            if last_pSS is not None and not np.all(last_pSS >= 0):
                last_pSS = None
        if last_pSS is None:
            last_pSS = self.get_steady_state_algebraic(max_iters=10, check_negative=False, set=False)
        last_flux = self.get_steady_state_target_flux(pSS=last_pSS, _set=False)
        for N in range(max_iters):
            iterated = inverse_iteration(matrix=T, guess=last_pSS, _solver=solver)
            last_pSS = iterated
            new_flux = self.get_steady_state_target_flux(pSS=last_pSS, _set=False)
            flux_change = new_flux - last_flux
            last_flux = new_flux
            if abs(flux_change) < last_flux * flux_fractional_convergence:
                break
            elif N == max_iters - 1 and not last_flux == 0:
                log.warning("Flux is nonzero and did not converge!")
        assert (last_pSS >= 0).all(), "Negative elements in pSS"
        assert last_flux >= 0, "Negative flux estimate from this pSS"
        self.pSS = last_pSS

    def get_steady_state_target_flux(self, pSS=None, _set=True):
        """`_analysis.py:317-384`"""
        T = self._T_csr()
        if not is_connected(T, self.indBasis, self.indTargets, directed=True):
            log.critical("There is no path in this matrix from the basis to the target, so no MFPT can be calculated.")
            return -1
        if pSS is None:
            pSS = np.squeeze(np.array(self.pSS))
        lagtime = self.tau * (self.n_lag + 1)
        not_targets = np.setdiff1d(range(self.nBins), self.indTargets)
        Jt = 0.0
        for jj in self.indTargets:
            col = np.asarray(T[:, int(jj)].todense()).ravel()
            Jt = Jt + np.sum(np.multiply(pSS[not_targets], col[not_targets]))
        if _set:
            self.lagtime = lagtime
            self.JtargetSS = Jt / self.lagtime
        else:
            return Jt / lagtime

    def get_committor(self, conv=1e-5):
        """`_analysis.py:527-606`: absorbing basis / target, iterate q <- T q until sum|dq| < conv."""
        F = self.fluxMatrix.copy()
        out = np.sum(F, 1)
        for i in range(F.shape[0]):
            if out[i] > 0:
                F[i, :] = F[i, :] / out[i]
            if out[i] == 0.0:
                F[i, i] = 1.0
        for ii in self.indBasis:
            F[ii, :] = 0.0
            F[ii, ii] = 1.0
        Fs = sp.csr_matrix(F)
        q = np.zeros((self.nBins, 1))
        q[self.indTargets, 0] = 1.0
        qp = np.ones_like(q)
        dconv = 100.0
        while dconv > conv:
            q[self.indTargets, 0] = 1.0
            q[self.indBasis, 0] = 0.0
            q = Fs @ q
            dconv = np.sum(np.abs(qp - q))
            qp = q.copy()
        self.q = q.squeeze()

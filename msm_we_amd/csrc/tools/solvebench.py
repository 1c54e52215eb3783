"""SURVEY §8d, solve timing: `get_steady_state` + `get_steady_state_target_flux` of this package (ARPACK / LAPACK / SuperLU on
the CSR matrix) against the reference's chain restated from _analysis.py:193-282 + utils.py:116-161 (dense `eig`, then inverse
iteration through an explicit sparse inverse), on row-stochastic matrices of K + 2 = 52, 194, 1002, 5002 states shaped like
cleaned haMSMs (local transitions + sparse long jumps + recycling from the target state to the basis state).  Host only."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from msm_we_amd.analysis import AnalysisMixin


def make_T(n, rng, band=12, extra=0.01):
    T = np.zeros((n, n))
    for i in range(n - 2):
        lo, hi = max(0, i - band), min(n - 2, i + band + 1)
        T[i, lo:hi] = rng.random(hi - lo)
        js = rng.integers(0, n, max(1, int(extra * n)))
        T[i, js] += rng.random(len(js)) * 0.05
        if i >= n - 2 - band:
            T[i, n - 1] += rng.random() * 0.2   # the last states reach the target
    T[n - 2, : band] = rng.random(band)          # basis state feeds the first states
    T[n - 2, n - 2] = 1.0
    T[n - 1, :] = 0.0
    T[n - 1, n - 2] = 1.0                          # target -> basis (recycling, _analysis.py:64-71)
    T /= T.sum(1, keepdims=True)
    return T


def reference_chain(T, indTargets, n_iter=10, conv=1e-4):
    t0 = time.perf_counter()
    vals, vecs = np.linalg.eig(T.T)                # _analysis.py:222
    p = np.real(vecs[:, np.argmax(np.real(vals))])
    p = p / p.sum()
    t_eig = time.perf_counter() - t0
    def flux(p):
        nt = np.setdiff1d(np.arange(len(p)), indTargets)
        return sum(np.sum(p[nt] * T[nt, j]) for j in indTargets)
    last = flux(p)
    t0 = time.perf_counter()
    n = len(p)
    for _ in range(n_iter):                        # utils.py:116-161: (T^T - I)^-1 formed explicitly
        inv = spla.inv(sp.csc_matrix(T.T - np.eye(n) * (1 - 1e-12)))
        p = inv @ p
        p = np.asarray(p).ravel()
        p = p / p.sum()
        new = flux(p)
        if abs(new - last) < new * conv:
            last = new
            break
        last = new
    return p, last, t_eig, time.perf_counter() - t0


class M(AnalysisMixin):
    pass


rng = np.random.default_rng(0)
print(f"{'states':>7} {'nnz':>9} | {'here: solve s':>13} | {'reference chain: eig s':>22} {'inverse it. s':>13} | max rel diff pSS, rel diff J")
for n in (52, 194, 1002, 5002):
    T = make_T(n, rng)
    m = M()
    m.Tmatrix = T
    m.Tmatrix_csr = sp.csr_matrix(T)
    m.indBasis, m.indTargets, m.nBins, m.tau, m.n_lag = np.array([n - 2]), np.array([n - 1]), n, 1.0, 0
    m.get_steady_state()                            # (first call pays library start-up)
    t0 = time.perf_counter()
    m.get_steady_state()
    m.get_steady_state_target_flux()
    t_here = time.perf_counter() - t0
    if n <= 1002 or os.environ.get("SOLVEBENCH_FULL"):
        p, J, t_eig, t_inv = reference_chain(T, m.indTargets)
        big = m.pSS > 1e-12 * m.pSS.max()
        d = np.max(np.abs(m.pSS[big] - p[big]) / p[big])
        print(f"{n:7d} {m.Tmatrix_csr.nnz:9d} | {t_here:13.4f} | {t_eig:22.3f} {t_inv:13.3f} | {d:.1e}, {abs(m.JtargetSS - J) / J:.1e}", flush=True)
    else:
        print(f"{n:7d} {m.Tmatrix_csr.nnz:9d} | {t_here:13.4f} | {'(not run: minutes)':>36} |", flush=True)

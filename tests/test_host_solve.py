"""Host-side solve helpers (msm_we_amd/analysis.py) against the definitions the reference uses (utils.py:87-161)."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.csgraph as csgraph

from msm_we_amd.analysis import _cpu_share, inverse_iteration, is_connected


def _reference_is_connected(matrix, sources, targets, directed=True):
    # the reference's predicate: no infinite shortest-path length from any source to any target
    d = csgraph.shortest_path(matrix, directed=directed, indices=sources)
    return np.inf not in d[:, targets]


def test_is_connected_equals_shortest_path_predicate():
    rng = np.random.default_rng(5)
    seen = set()
    for trial in range(200):
        n = int(rng.integers(3, 40))
        A = sp.csr_matrix(rng.random((n, n)) * (rng.random((n, n)) < rng.choice([0.03, 0.08, 0.2])))
        src = rng.choice(n, size=int(rng.integers(1, 3)), replace=False)
        tgt = rng.choice(n, size=int(rng.integers(1, 3)), replace=False)
        for directed in (True, False):
            want = bool(_reference_is_connected(A, src, tgt, directed))
            assert is_connected(A, src, tgt, directed=directed) == want
            seen.add(want)
    assert seen == {True, False}


def test_inverse_iteration_dense_and_sparse_agree():
    rng = np.random.default_rng(11)
    n = 300
    for density in (0.03, 0.5):      # SuperLU below 10 % density, LAPACK above
        T = rng.random((n, n)) * (rng.random((n, n)) < density) + np.eye(n) * 0.1
        T /= T.sum(1, keepdims=True)
        vals, vecs = np.linalg.eig(T.T)
        pi = np.real(vecs[:, np.argmax(np.real(vals))])
        pi /= pi.sum()
        x = inverse_iteration(np.full(n, 1.0 / n), sp.csr_matrix(T))
        np.testing.assert_allclose(x, pi, rtol=1e-8, atol=1e-14)


def test_dense_lu_is_chosen_for_patterns_that_fill_in():
    """banded pattern -> sparse LU; a percent of long-range entries (what cleaned haMSMs have) -> dense LU; both give the
    stationary vector"""
    from msm_we_amd.analysis import _prefer_dense_lu

    rng = np.random.default_rng(2)
    n = 400
    band = sp.diags([rng.random(n - abs(k)) for k in range(-5, 6)], list(range(-5, 6))).tocsr()
    assert not _prefer_dense_lu(band, n)
    T = band.toarray() + (rng.random((n, n)) < 0.01) * rng.random((n, n))
    T /= T.sum(1, keepdims=True)
    Ts = sp.csr_matrix(T)
    assert _prefer_dense_lu(Ts, n)
    vals, vecs = np.linalg.eig(T.T)
    pi = np.real(vecs[:, np.argmax(np.real(vals))])
    pi /= pi.sum()
    np.testing.assert_allclose(inverse_iteration(np.full(n, 1.0 / n), Ts), pi, rtol=1e-8, atol=1e-14)


def test_cpu_share_is_sane():
    import os
    assert 1 <= _cpu_share() <= (os.cpu_count() or 1)

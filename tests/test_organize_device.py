"""a12 / a13 without the host round trip: the primitives that keep the flux matrix in HBM from get_fluxMatrix to get_Tmatrix
(`hamsm_pattern_csr`, `hamsm_np_sum`, `hamsm_permute_scale`, the count-pass form of `hamsm_tmatrix_csr`)."""
import numpy as np
import pytest
import scipy.sparse as sp

SIZES = [3, 9, 34, 91, 130, 629, 1002]


def _matrix(rng, n, density=0.3):
    a = rng.random((n, n)) * 10.0 ** rng.integers(-6, 3, size=(n, n))
    a[rng.random((n, n)) > density] = 0.0
    return a


def _check(h, n, seed):
    rng = np.random.default_rng(seed)
    a = _matrix(rng, n)
    d_a = h.to_device(a)
    # np.sum of the array as numpy lays it out, C order ...
    assert h.np_sum(d_a, n * n) == np.sum(a)
    # ... and of `a[perm, :][:, perm]` (column-major in numpy): the organize step's divisor
    perm = rng.permutation(n)
    p = a[perm, :][:, perm]
    d_t, d_p = h.alloc((n, n), np.float64), h.alloc((n, n), np.float64)
    h.permute_scale(d_a, n, perm, d_t, 1.0, transpose=True)
    total = h.np_sum(d_t, n * n)
    assert total == np.sum(p)
    h.permute_scale(d_a, n, perm, d_p, total)
    assert np.array_equal(d_p.to_host(), p / np.sum(p))
    # sparsity pattern
    indptr, indices = h.pattern_csr(d_a, n)
    ref = sp.csr_matrix(a != 0)
    assert np.array_equal(indptr, ref.indptr) and np.array_equal(indices, ref.indices)
    # T as CSR, sized by the count pass; all-zero matrix: self loops
    basis, target = [n - 2], [n - 1]
    ip, ix, dat = h.tmatrix_csr(d_a, n, basis, target)
    T = sp.csr_matrix((dat, ix, ip), shape=(n, n)).toarray()
    out = a.sum(1)
    ref_T = np.where(out[:, None] > 0, a / np.where(out > 0, out, 1)[:, None], 0.0)
    for i in np.nonzero(out == 0)[0]:
        ref_T[i, i] = 1.0
    ref_T[target[0]] = 0.0
    ref_T[target[0], basis[0]] = 1.0
    np.testing.assert_allclose(T, ref_T, rtol=1e-15, atol=0)
    assert len(dat) == np.count_nonzero(ref_T)


@pytest.mark.parametrize("n", SIZES)
def test_organize_primitives_oracle(oracle_h, n):
    _check(oracle_h, n, 100 + n)


@pytest.mark.gpu
@pytest.mark.parametrize("n", SIZES + [2606])
def test_organize_primitives_hip(hip_h, n):
    _check(hip_h, n, 100 + n)


def test_np_sum_long_runs_oracle(oracle_h):
    rng = np.random.default_rng(5)
    for n in (1, 7, 8, 127, 128, 129, 8191, 8192, 8193, 50001):
        v = rng.normal(size=n) * 10.0 ** rng.integers(-8, 8, size=n)
        assert oracle_h.np_sum(oracle_h.to_device(v), n) == np.sum(v), n


@pytest.mark.gpu
def test_np_sum_long_runs_hip(hip_h):
    rng = np.random.default_rng(5)
    for n in (1, 7, 8, 127, 128, 129, 8191, 8192, 8193, 50001, 5002 * 5002):
        v = rng.normal(size=n) * 10.0 ** rng.integers(-8, 8, size=n)
        assert hip_h.np_sum(hip_h.to_device(v), n) == np.sum(v), n

"""Streaming PCA fit by covariance accumulation (`hamsm_pca_begin / accumulate / finish`, opt-in
`modelWE.pca_fit = "device"`): the oracle's sums against extended-precision NumPy, the HIP library's against the oracle's bit
for bit, the components against scikit-learn's exact PCA, and the model-level fit against the reference's own IncrementalPCA
(golden c1p_d30_pca: same number of components, the same subspace to a small angle -- not the same bits: IncrementalPCA with
a component cap truncates at every batch)."""
import numpy as np
import pytest

from conftest import golden_iterations, load_golden
from msm_we_amd import _abi
from msm_we_amd.pca import CovariancePCA


def _blocks(seed, D, sizes, dtype):
    rng = np.random.default_rng(seed)
    mix = rng.normal(size=(D, D)) * (rng.uniform(size=D) < 0.4)[:, None] + np.diag(rng.uniform(0.05, 3.0, D))
    return [((rng.normal(size=(n, D)) @ mix) + rng.normal(size=D) * 50).astype(dtype) for n in sizes]


def _sums(h, blocks, ldx_pad=0):
    D = blocks[0].shape[1]
    h.pca_begin(D)
    for X in blocks:
        n = len(X)
        if ldx_pad:
            Xp = np.full((n, D + ldx_pad), np.nan, dtype=X.dtype)
            Xp[:, :D] = X
            X = Xp
        d = h.to_device(X)
        h.pca_accumulate(d, _abi.F32 if X.dtype == np.float32 else _abi.F64, D + ldx_pad, n)
        h.sync()
        d.free()
    return h.pca_finish()


SHAPES = [(30, [128] * 5, np.float32), (13, [4001], np.float64), (31, [7, 1, 2500], np.float32), (32, [3000, 5], np.float64),
          (200, [1500, 900], np.float32), (3, [2], np.float64), (64, [70000], np.float32)]


@pytest.mark.parametrize("D,sizes,dtype", SHAPES)
def test_oracle_sums_match_extended_precision(oracle_h, D, sizes, dtype):
    blocks = _blocks(5, D, sizes, dtype)
    n, shift, S1, S2 = _sums(oracle_h, blocks, ldx_pad=3)
    X = np.concatenate(blocks).astype(np.longdouble)
    assert n == len(X) and np.array_equal(shift, np.concatenate(blocks)[0].astype(np.float64))
    Z = X - shift.astype(np.longdouble)
    r1, r2 = Z.sum(axis=0), Z.T @ Z
    scale = np.sqrt(np.outer(np.diag(r2), np.diag(r2))).astype(np.float64) + 1e-300
    assert np.max(np.abs(S2 - r2.astype(np.float64)) / scale) < 1e-12
    assert np.allclose(S1, r1.astype(np.float64), rtol=1e-12, atol=1e-9 * np.sqrt(np.diag(r2).astype(np.float64)).max())
    assert np.array_equal(S2, S2.T)


def test_fit_matches_sklearn_exact_pca(oracle_h):
    from sklearn.decomposition import PCA

    blocks = _blocks(9, 24, [500, 300, 1200], np.float32)
    X = np.concatenate(blocks).astype(np.float64)
    ref = PCA(n_components=7, svd_solver="full").fit(X)
    mine = CovariancePCA(n_components=7).fit_from_sums(*_sums(oracle_h, blocks))
    assert mine.n_samples_seen_ == len(X) and mine.n_components_ == 7
    np.testing.assert_allclose(mine.mean_, ref.mean_, rtol=1e-12)
    np.testing.assert_allclose(mine.explained_variance_, ref.explained_variance_, rtol=1e-10)
    np.testing.assert_allclose(mine.explained_variance_ratio_, ref.explained_variance_ratio_, rtol=1e-10)
    np.testing.assert_allclose(mine.singular_values_, ref.singular_values_, rtol=1e-10)
    np.testing.assert_allclose(mine.noise_variance_, ref.noise_variance_, rtol=1e-9)
    for a, b in zip(mine.components_, ref.components_):
        assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 1e-9          # (sign conventions differ between estimators)
        assert a[np.argmax(np.abs(a))] > 0                                   # IncrementalPCA's: svd_flip(u_based_decision=False)
    np.testing.assert_allclose(np.abs(mine.transform(X[:50])), np.abs(ref.transform(X[:50])), atol=1e-8)
    np.testing.assert_allclose(mine.inverse_transform(mine.transform(X[:5])), ref.inverse_transform(ref.transform(X[:5])), atol=1e-8)


def test_bad_calls_are_refused(oracle_h):
    h = oracle_h
    with pytest.raises(_abi.HamsmError):
        h.lib.hamsm_pca_accumulate.restype  # noqa: B018  (bound)
        h.pca_accumulate(h.alloc(4, np.float64), _abi.F64, 2, 2)            # before begin
    h.pca_begin(4)
    d = h.to_device(np.zeros((3, 4)))
    with pytest.raises(_abi.HamsmError):
        h.pca_accumulate(d, _abi.F64, 3, 3)                                  # ldx < D
    with pytest.raises(_abi.HamsmError):
        h.pca_accumulate(d, 7, 4, 3)                                         # dtype
    h.pca_accumulate(d, _abi.F64, 4, 0)                                      # empty block: nothing
    n, shift, S1, S2 = h.pca_finish()
    assert n == 0 and not S2.any()
    with pytest.raises(ValueError):
        CovariancePCA().fit_from_sums(n, shift, S1, S2)


def _device_fit_model(h, g, meta):
    from msm_we_amd.data import InMemorySource
    from msm_we_amd.model import modelWE

    m = modelWE(h=h)
    m.pca_fit = "device"
    m.set_iteration_source(InMemorySource(golden_iterations(g)))
    m.initialize(["golden.h5"], {"coords": None, "nAtoms": meta["D"], "coord_ndim": 1}, "golden", basis_pcoord_bounds=meta["basis"],
                 target_pcoord_bounds=meta["target"], tau=meta["tau"], dim_reduce_method="pca")
    m.get_iterations()
    m.get_coordSet(m.maxIter)
    m.dimReduce()
    return m


def check_model_level_fit(h):
    """`pca_fit = "device"` on the inputs of golden c1p_d30_pca: as many components as the reference kept, spanning the same
    subspace as its IncrementalPCA to a small angle; the projected frames (staging) are those of `transform`"""
    g, meta = load_golden("c1p_d30_pca")
    m = _device_fit_model(h, g, meta)
    c = m.coordinates
    assert type(c).__name__ == "CovariancePCA" and m.ndim == c.n_components_ == int(g["pca_ndim"])
    ref_W = np.asarray(g["pca_components"])
    s = np.linalg.svd(c.components_ @ ref_W.T, compute_uv=False)          # cosines of the principal angles
    assert s.min() > 0.97, s
    np.testing.assert_allclose(c.mean_, g["pca_mean"], rtol=1e-9, atol=1e-9)
    store = m.stage_iterations()
    it = 5
    f0, f1 = store.span(it, it)
    Xc = golden_iterations(g)[it]["Xc"]
    np.testing.assert_allclose(store.Xc.to_host().reshape(-1, m.ndim)[f0:f1], c.transform(Xc), rtol=1e-9, atol=1e-9)
    return m


def test_model_level_device_fit_against_the_reference_fit(oracle_h):
    m = check_model_level_fit(oracle_h)
    import pickle

    c2 = pickle.loads(pickle.dumps(m.coordinates))
    assert np.array_equal(c2.components_, m.coordinates.components_)


@pytest.mark.gpu
@pytest.mark.parametrize("D,sizes,dtype", SHAPES + [(816, [3000], np.float32), (129, [100000, 17], np.float64)])
def test_hip_sums_equal_oracle_sums(hip_h, oracle_h, D, sizes, dtype):
    blocks = _blocks(5, D, sizes, dtype)
    a = _sums(oracle_h, blocks, ldx_pad=1)
    b = _sums(hip_h, blocks, ldx_pad=1)
    assert a[0] == b[0]
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)


@pytest.mark.gpu
def test_model_level_device_fit_on_gpu(hip_h, oracle_h):
    m = check_model_level_fit(hip_h)
    g, meta = load_golden("c1p_d30_pca")
    o = _device_fit_model(oracle_h, g, meta)
    assert np.array_equal(m.coordinates.components_, o.coordinates.components_) and np.array_equal(m.coordinates.mean_, o.coordinates.mean_)

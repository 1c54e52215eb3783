"""a2: the dimensionality-reduction transform (`hamsm_project`).  The oracle against scikit-learn's own
`PCA.transform` / `IncrementalPCA.transform` (the objects the reference stores in `model.coordinates`,
_dimensionality.py:161-241); the HIP kernel against the oracle bit for bit."""
import numpy as np
import pytest

sk = pytest.importorskip("sklearn.decomposition")


def _data(rng, n, Draw, dtype):
    B = rng.normal(size=(Draw, Draw)) / np.sqrt(Draw)
    return ((rng.normal(size=(n, Draw)) * np.linspace(3, 0.1, Draw)) @ B + rng.normal(size=Draw) * 5).astype(dtype)


@pytest.mark.parametrize("cls,whiten,dtype", [("PCA", False, np.float32), ("PCA", True, np.float64),
                                              ("IncrementalPCA", False, np.float32)])
def test_oracle_projection_is_sklearn_transform(oracle_h, cls, whiten, dtype):
    rng = np.random.default_rng(0)
    X = _data(rng, 700, 45, dtype)
    kw = dict(n_components=7, whiten=whiten)
    pca = getattr(sk, cls)(**kw).fit(X[:400].astype(np.float64))
    ref = pca.transform(X)
    h = oracle_h
    scale = None
    if whiten:
        scale = np.sqrt(pca.explained_variance_)  # _base.py: X_transformed /= sqrt(explained_variance_) (clipped from below)
    d_X = h.to_device(X)
    Y = h.project(d_X, 0 if dtype == np.float32 else 1, X.shape[1], len(X), pca.components_, pca.mean_, scale).to_host().reshape(len(X), -1)
    # BLAS blocks and vectorises its sums; ours are plain chains: agreement to a few ulp of the largest term
    np.testing.assert_allclose(Y, ref, rtol=1e-11, atol=1e-11 * np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("n,Draw,d,dtype,mean,whiten", [(1000, 30, 5, np.float32, True, False), (4097, 301, 33, np.float32, True, True),
                                                        (513, 64, 16, np.float64, False, False), (70, 3, 2, np.float32, True, False),
                                                        (20000, 900, 70, np.float32, True, False)])
def test_projection_matches_oracle(hip_h, oracle_h, n, Draw, d, dtype, mean, whiten):
    out = []
    for h in (oracle_h, hip_h):
        rng = np.random.default_rng(3)
        X = _data(rng, n, Draw, dtype)
        W = rng.normal(size=(d, Draw))
        mu = rng.normal(size=Draw) if mean else None
        sc = rng.uniform(0.5, 2.0, size=d) if whiten else None
        ld = Draw + 3  # padded rows
        Xp = np.zeros((n, ld), dtype=dtype)
        Xp[:, :Draw] = X
        d_Y = h.project(h.to_device(Xp), 0 if dtype == np.float32 else 1, ld, n, W, mu, sc)
        out.append(d_Y.to_host().reshape(n, d))
    assert np.array_equal(out[0], out[1])

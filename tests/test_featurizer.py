"""a1: the distance-to-reference featurizer of the reference's tests (tests/test_msm_we.py:19-36) as an entry point of the
C ABI.  MDAnalysis is not in this image, so the bar is the published algorithm (float32 positions and difference, float64
squares / sum / root), restated here in NumPy independently of the oracle's C."""
import numpy as np
import pytest

from msm_we_amd import _abi


def _numpy_dist(coords, idx, ref):
    c32 = np.asarray(coords).astype(np.float32)[:, idx, :]
    d = (c32 - ref[None].astype(np.float32)).astype(np.float64)          # float32 difference, then widened
    rsq = d[..., 0] * d[..., 0]
    rsq = rsq + d[..., 1] * d[..., 1]
    rsq = rsq + d[..., 2] * d[..., 2]
    return np.sqrt(rsq)


def _case(rng, n, n_atoms, n_sel, dtype):
    coords = (rng.normal(size=(n, n_atoms, 3)) * 10).astype(dtype)
    idx = np.sort(rng.choice(n_atoms, n_sel, replace=False)).astype(np.int32)
    ref = (rng.normal(size=(n_sel, 3)) * 10).astype(np.float32)
    return coords, idx, ref


@pytest.mark.parametrize("n,n_atoms,n_sel,dtype", [(37, 272, 80, np.float32), (5, 9, 9, np.float64), (1, 3, 1, np.float32)])
def test_oracle_featurizer_is_the_published_algorithm(oracle_h, n, n_atoms, n_sel, dtype):
    rng = np.random.default_rng(3)
    coords, idx, ref = _case(rng, n, n_atoms, n_sel, dtype)
    h = oracle_h
    d_c = h.to_device(coords.reshape(n, -1))
    X = h.featurize_dist(d_c, _abi.F32 if dtype == np.float32 else _abi.F64, 3 * n_atoms, n, n_atoms, idx, ref).to_host().reshape(n, n_sel)
    assert np.array_equal(X, _numpy_dist(coords, idx, ref))
    with pytest.raises(_abi.HamsmError):
        h.featurize_dist(d_c, _abi.F32, 3 * n_atoms, n, n_atoms, np.array([n_atoms], np.int32), ref[:1])


def test_model_with_device_featurizer_equals_host_featurizer(oracle_h):
    """staging raw coordinates + `set_distance_featurizer` gives the model that a host `processCoordinates` computing the same
    distances gives (labels, flux matrix)"""
    from conftest import RectilinearBinMapper
    from msm_we_amd.data import InMemorySource
    from msm_we_amd.model import modelWE

    rng = np.random.default_rng(11)
    n_atoms, n_sel, S, n_iters, segs = 12, 7, 2, 12, 96
    idx = np.sort(rng.choice(n_atoms, n_sel, replace=False)).astype(np.int32)
    ref = rng.normal(size=(n_sel, 3)).astype(np.float32)
    iters = {}
    for it in range(1, n_iters + 1):
        p0 = rng.uniform(0, S, size=(segs, 1))
        p1 = np.clip(p0 + rng.normal(scale=0.3, size=(segs, 1)), 0, S + 0.4)
        mk = lambda p: (rng.normal(size=(segs, n_atoms, 3)) + 3 * np.floor(p)[:, :, None]).astype(np.float32)
        w = rng.exponential(size=segs)
        iters[it] = {"parent": mk(p0), "child": mk(p1), "p0": p0, "p1": p1, "w": w / w.sum()}

    def build(device):
        m = modelWE(h=oracle_h)
        m.set_iteration_source(InMemorySource(iters))
        m.initialize(["x.h5"], {"coords": None, "nAtoms": n_atoms, "coord_ndim": 3}, "feat", basis_pcoord_bounds=[[-np.inf, 0.05]],
                     target_pcoord_bounds=[[S - 0.05, np.inf]], tau=1.0, pcoord_ndim=1)
        if device:
            m.set_distance_featurizer(idx, ref)
        else:
            m.processCoordinates = lambda coords: _numpy_dist(np.asarray(coords).reshape(len(coords), n_atoms, 3), idx, ref)
        m.get_iterations()
        m.get_coordSet(m.maxIter)
        m.dimReduce()
        b = np.arange(S + 1, dtype=np.float32)
        b[-1] = np.inf
        m.cluster_coordinates(6, streaming=True, use_ray=True, stratified=True, user_bin_mapper=RectilinearBinMapper([b]),
                              random_state=5)
        m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter, use_ray=True)
        return m

    a, b = build(True), build(False)
    assert a._store.D == n_sel
    assert np.array_equal(np.concatenate(a.dtrajs), np.concatenate(b.dtrajs))
    assert np.array_equal(a.fluxMatrixRaw, b.fluxMatrixRaw)


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_atoms,n_sel,dtype", [(4099, 272, 80, np.float32), (513, 40, 40, np.float64), (3, 5, 2, np.float32)])
def test_hip_featurizer_matches_oracle(hip_h, oracle_h, n, n_atoms, n_sel, dtype):
    rng = np.random.default_rng(4)
    coords, idx, ref = _case(rng, n, n_atoms, n_sel, dtype)
    coords[0, idx[0]] = ref[0]                       # a zero distance
    outs = []
    for h in (oracle_h, hip_h):
        d_c = h.to_device(coords.reshape(n, -1))
        outs.append(h.featurize_dist(d_c, _abi.F32 if dtype == np.float32 else _abi.F64, 3 * n_atoms, n, n_atoms, idx, ref).to_host())
    assert np.array_equal(outs[0], outs[1])
    assert outs[1].reshape(n, n_sel)[0, 0] == 0.0

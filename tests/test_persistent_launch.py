"""(An experiment that is off and not part of the shipped library, DESIGN.md 5.3a: csrc/experiments/train_persist.inc.)"""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.experiments]   # `make EXPERIMENTS=1`, `pytest -m experiments`


@pytest.mark.parametrize("shape", ["c3_like", "reseeding", "weighted_f64", "many_strata"])
def test_persistent_training_launch_equals_multikernel_path(exp_lib, monkeypatch, shape):
    """HAMSM_TRAIN_PERSIST=1: the whole planned pass as ONE persistent launch (a stratum's chain of E-steps and M-steps on the
    workgroups of one XCD, XCD-local barriers).  Off by default because it is slower; its results must be
    those of the multi-kernel path bit for bit: centroids, counts, bookkeeping, labels -- including chains that stop and
    resume for re-seeding, sample weights with float64 frames, and more strata than XCDs."""
    from conftest import RectilinearBinMapper
    from msm_we_amd import _abi, synthetic
    from msm_we_amd.data import DeviceStore
    from msm_we_amd.model import modelWE

    cfg, dtype, weights = {
        "c3_like": (synthetic.WEConfig("p1", 3, 30, 6000, 30, 4, 250), np.float32, False),
        "reseeding": (synthetic.WEConfig("p2", 5, 30, 12000, 64, 4, 600), np.float32, False),
        "weighted_f64": (synthetic.WEConfig("p3", 7, 25, 3000, 13, 3, 40), np.float64, True),
        "many_strata": (synthetic.WEConfig("p4", 11, 25, 6000, 10, 11, 24), np.float32, False),
    }[shape]
    blk = synthetic.make_block(cfg, 1, cfg.n_iters)
    out = []
    for persist in ("0", "1"):
        monkeypatch.setenv("HAMSM_TRAIN_PERSIST", persist)
        h = _abi.Hamsm(lib=exp_lib)
        try:
            store = DeviceStore.from_host(h, blk["Xp"].astype(dtype), blk["Xc"].astype(dtype), blk["p0"], blk["p1"], blk["w"], blk["offsets"])
            m = modelWE(h=h)
            m.pcoord_ndim = 1
            basis, target = synthetic.basis_target_bounds(cfg.S)
            m.basis_pcoord_bounds, m.target_pcoord_bounds, m.tau = basis, target, 1.0
            m.use_weights_in_clustering = weights
            m.attach_store(store)
            m.maxIter = cfg.n_iters + 1
            m.numSegments = np.diff(blk["offsets"]).astype(float)
            bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
            m.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=3)
            assert m.training_path == "planned"
            out.append(([cm.cluster_centers_.copy() for cm in m.clusters.cluster_models],
                        [cm._counts.copy() for cm in m.clusters.cluster_models],
                        [(cm._n_since_last_reassign, cm.n_steps_) for cm in m.clusters.cluster_models], np.concatenate(m.dtrajs)))
        finally:
            h.close()
    for a, b in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert np.array_equal(a, b)
    assert out[0][2] == out[1][2] and np.array_equal(out[0][3], out[1][3])



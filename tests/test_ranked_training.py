"""The ranked E-step of MFMA-bound training calls (k_step_assign_ranked, train.hip): centroids ranked on the bf16
matrix unit from fragments the M-step maintains, a label final only under a proven bound, rows between two candidates settled by
the contract's two f64 chains, everything else swept exactly.  The labels are the contract's by construction, so a whole planned
pass -- seeding, steps, re-seeding on the device, stop / settle / resume -- must stay bit-identical to the oracle with this path
FORCED on shapes of every kind (production takes it by size: >= 0.15 GFLOP per call, BASELINE config 5's regime).
Arithmetic it must reproduce: sklearn `_k_means_lloyd.pyx:168-218` as called from `/root/reference/msm_we/_hamsm/_clustering.py:909-911`."""
import ctypes

import numpy as np
import pytest

from msm_we_amd import synthetic
from test_training_paths import _run

pytestmark = pytest.mark.gpu


def _rank(h, mode=-1, reset=True):
    """(certified, settled between two candidates, swept exactly, runs) since the last reset; mode >= 0 sets HAMSM_STEP_RANKED"""
    out = (ctypes.c_ulonglong * 4)()
    fn = h.lib.hamsm_debug_rank_stats
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    fn.restype = ctypes.c_int
    assert fn(h.ctx, out, int(reset), int(mode)) == 0
    return dict(certified=out[0], two=out[1], full=out[2], runs=out[3])


CASES = [
    # name, config (seed, iterations, segments, D, S, K_s), clusters, kwargs
    ("balanced", synthetic.WEConfig("rk1", 3, 30, 1500, 10, 3, 12), 12, {}),
    ("c3-like", synthetic.WEConfig("rk2", 8, 24, 4000, 30, 4, 60), 60, {}),
    ("weighted", synthetic.WEConfig("rk3", 4, 30, 1200, 10, 2, 12), 12, dict(weights=True)),
    ("d64", synthetic.WEConfig("rk5", 12, 20, 2400, 64, 3, 40), 40, {}),
    ("d48-two-tiles-per-wave", synthetic.WEConfig("rk5b", 14, 8, 9000, 48, 2, 300), 300, {}),
    ("d17", synthetic.WEConfig("rk5c", 15, 12, 2000, 17, 2, 33), 33, {}),
    ("many-strata", synthetic.WEConfig("rk6", 13, 20, 3300, 12, 11, 8), 8, {}),
    ("more-centroids-than-modes", synthetic.WEConfig("rk7", 5, 20, 2500, 6, 2, 2), 20, {}),   # twins: rows between two candidates
    ("reseeding", synthetic.WEConfig("rk8", 5, 16, 2500, 6, 2, 2), 20, dict(shift=True)),      # device re-seeding keeps the fragments
    ("reseeding-weighted", synthetic.WEConfig("rk8w", 5, 16, 2500, 6, 2, 2), 20, dict(shift=True, weights=True)),
    ("outliers", synthetic.WEConfig("rk9", 5, 12, 2500, 6, 2, 2), 40, dict(outliers=True)),   # scale 100 vs 0.01: wide bounds
    # more than 32 launch groups per run: the enqueue runs a bounded distance ahead of the device, polls the chains' stop words,
    # drops stopped strata from the launches that follow and ends a run early (run_steps, "bounded run-ahead") -- with chains
    # that do stop (re-seeding), on two queues and on one (11 strata)
    ("reseeding-long", synthetic.WEConfig("rk8l", 5, 70, 2500, 6, 2, 2), 20, dict(shift=True)),
    ("reseeding-weighted-long", synthetic.WEConfig("rk8wl", 6, 50, 2000, 6, 3, 2), 16, dict(shift=True, weights=True)),
    ("many-strata-long", synthetic.WEConfig("rk6l", 13, 45, 3300, 12, 11, 8), 8, dict(shift=True)),
]


def _shape(h, wide, direct):
    """(test hook) 256-row E-step workgroups never / by size / always; member rows from the frames (1) or the f64 batch copy (0)"""
    fn = h.lib.hamsm_debug_step_shape
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    fn.restype = ctypes.c_int
    assert fn(h.ctx, int(wide), int(direct)) == 0


# round 5: the M-step reads member rows from the frames through the row list (no f64 batch copy), and the E-step takes 256-row
# workgroups where 128-row ones overfill the chip (C5's shape) -- every case runs in the shipped shape, with the wide workgroups
# FORCED, and in round 4's shape (batch copy, 128 rows)
@pytest.mark.parametrize("shape", ["default", "wide", "four-waves", "copy"])
@pytest.mark.parametrize("name,cfg,k,kw", CASES, ids=[c[0] for c in CASES])
def test_ranked_training_pass_equals_the_oracle(hip_h, oracle_h, name, cfg, k, kw, shape):
    o = _run(oracle_h, cfg, k, True, **kw)
    _rank(hip_h, 2)
    _shape(hip_h, *{"default": (1, 1), "wide": (2, 1), "four-waves": (3, 1), "copy": (0, 0)}[shape])
    g = _run(hip_h, cfg, k, True, **kw)
    st = _rank(hip_h, 1)
    assert st["runs"] >= 1 and st["certified"] + st["two"] + st["full"] > 0, st
    assert g[0].training_path == o[0].training_path
    assert np.array_equal(g[1], o[1]), f"{int((g[1] != o[1]).sum())} labels differ"
    for i, (x, y) in enumerate(zip(g[2] + g[3], o[2] + o[3])):
        assert np.array_equal(x, y), f"array {i} (centroid tables, then counts) differs in {int((x != y).sum())} places"
    # the shipped default for calls this small is the f64 E-step: same results, and the ranked path is not taken
    g2 = _run(hip_h, cfg, k, True, **kw)
    assert _rank(hip_h)["runs"] == 0
    assert np.array_equal(g2[1], g[1])
    for x, y in zip(g2[2] + g2[3], g[2] + g[3]):
        assert np.array_equal(x, y)


def test_ranked_pass_with_adversarial_rows(hip_h, oracle_h):
    """exact ties (duplicated rows become duplicated seeds: every member of such a cluster ties between twins), rows scaled by
    1e18 (the f32 scores overflow: nothing may be certified), NaN / inf rows (label 0 like the oracle's all-NaN rule): the
    exact parts of the kernel decide all of these"""
    from conftest import RectilinearBinMapper
    from msm_we_amd.data import InMemorySource
    from msm_we_amd.model import modelWE

    cfg = synthetic.WEConfig("rka", 31, 10, 1500, 12, 2, 16)
    lat = synthetic.latent_centres(cfg)
    iters = {i: synthetic.make_iteration(cfg, i, lat) for i in range(1, cfg.n_iters + 1)}
    rng = np.random.default_rng(5)
    for it, d in iters.items():
        for key in ("Xp", "Xc"):
            X = d[key]
            X[rng.choice(len(X), 40, replace=False)] = X[0]            # many copies of one row
            X[rng.choice(len(X), 5, replace=False)] *= np.float32(1e18)
            if it > 3:   # (a NaN in the seeding batch would poison k-means++ itself, in the reference too)
                X[rng.choice(len(X), 3, replace=False), 2] = np.nan
                X[rng.choice(len(X), 2, replace=False), 5] = np.inf
    res = []
    for h in (oracle_h, hip_h):
        if h is hip_h:
            _rank(h, 2)
        m = modelWE(h=h)
        m.set_iteration_source(InMemorySource(iters))
        basis, target = synthetic.basis_target_bounds(cfg.S)
        m.initialize(["x"], {"coords": None, "nAtoms": cfg.D, "coord_ndim": 1}, "t", basis_pcoord_bounds=basis,
                     target_pcoord_bounds=target, tau=1.0)
        m.get_iterations()
        m.dimReduce()
        bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
        try:
            m.cluster_coordinates(16, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=3)
            res.append(("ok", [cm.cluster_centers_.copy() for cm in m.clusters.cluster_models],
                        [cm._counts.copy() for cm in m.clusters.cluster_models], np.concatenate(m.dtrajs)))
        except Exception as e:  # noqa: BLE001 -- whatever the oracle does with such data, the HIP path must do the same
            res.append((type(e).__name__,))
    st = _rank(hip_h, 1)
    assert res[0][0] == res[1][0]
    if res[0][0] == "ok":
        assert st["runs"] >= 1 and st["full"] > 0
        for x, y in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
            assert np.array_equal(x, y, equal_nan=True)
        assert np.array_equal(res[0][3], res[1][3])


def test_c5_shape_takes_the_ranked_path_by_size(hip_lib, oracle_lib):
    """BASELINE config 5's shape (50 000 segments per iteration, 64 features, 4 x 1250): the shipped default takes the ranked
    E-step there (>= 0.15 GFLOP per call); 3 iterations against the oracle, bit for bit, and most rows certified"""
    from conftest import RectilinearBinMapper
    from msm_we_amd import _abi
    from msm_we_amd.data import DeviceStore
    from msm_we_amd.model import modelWE

    cfg = synthetic.WEConfig("c5r", 5, 3, 50000, 64, 4, 1250)
    blk = synthetic.make_block(cfg, 1, cfg.n_iters)
    out = []
    for lib in (oracle_lib, hip_lib):
        h = _abi.Hamsm(lib=lib)
        try:
            store = DeviceStore.from_host(h, blk["Xp"], blk["Xc"], blk["p0"], blk["p1"], blk["w"], blk["offsets"])
            m = modelWE(h=h)
            m.pcoord_ndim = 1
            basis, target = synthetic.basis_target_bounds(cfg.S)
            m.basis_pcoord_bounds, m.target_pcoord_bounds, m.tau = basis, target, 1.0
            m.attach_store(store)
            m.maxIter = cfg.n_iters + 1
            m.numSegments = np.diff(blk["offsets"]).astype(float)
            m.launch_ray_discretization = lambda progress_bar=None: None   # training only (the oracle's discretization would take minutes)
            bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
            m.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=5)
            out.append(([cm.cluster_centers_.copy() for cm in m.clusters.cluster_models], [cm._counts.copy() for cm in m.clusters.cluster_models]))
            if lib is hip_lib:
                st = _rank(h)
                assert st["runs"] >= 1, st
                tot = st["certified"] + st["two"] + st["full"]
                assert tot > 0 and st["full"] < 0.02 * tot, st
        finally:
            h.close()
    for x, y in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert np.array_equal(x, y)


def _plain(h, plain, lrows):
    """(test hook) the f64-MFMA training kernels: member rows from the frames (1) or the f64 batch copy (0); row list in LDS"""
    fn = h.lib.hamsm_debug_step_plain
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    fn.restype = ctypes.c_int
    assert fn(h.ctx, int(plain), int(lrows)) == 0


PLAIN_CASES = [c for c in CASES if c[0] in ("balanced", "c3-like", "weighted", "d64", "d17", "many-strata", "reseeding-weighted", "outliers",
                                            "reseeding-long")] + [
    ("d100-two-registers-per-lane", synthetic.WEConfig("pl1", 21, 12, 2000, 100, 2, 20), 20, {}),
    ("batches-beyond-4096-rows", synthetic.WEConfig("pl2", 22, 6, 11000, 12, 2, 30), 30, {}),          # UPD_NBK = 4: no row list in LDS
    ("batches-beyond-4096-rows-weighted", synthetic.WEConfig("pl3", 23, 6, 11000, 12, 2, 30), 30, dict(weights=True)),
]


# end of round 5: the f64-MFMA kernels (calls too small for the ranked E-step: C3's) drop the f64 batch copy too -- the M-step takes
# member rows from the frames, the call's row list waits in LDS (batches of <= 4096 rows) -- C3 training 17.8 -> 16.95 ms.  Shipped
# shape, without the LDS row list, and round 4's shape (copy) against the oracle.
@pytest.mark.parametrize("shape", ["default", "frames", "copy"])
@pytest.mark.parametrize("name,cfg,k,kw", PLAIN_CASES, ids=[c[0] for c in PLAIN_CASES])
def test_plain_training_pass_equals_the_oracle_in_every_shape(hip_h, oracle_h, name, cfg, k, kw, shape):
    o = _run(oracle_h, cfg, k, True, **kw)
    _rank(hip_h, 0)   # never the ranked E-step
    _plain(hip_h, *{"default": (1, 1), "frames": (1, 0), "copy": (0, 0)}[shape])
    try:
        g = _run(hip_h, cfg, k, True, **kw)
    finally:
        _plain(hip_h, 1, 1)
        st = _rank(hip_h, 1)
    assert st["runs"] == 0
    assert g[0].training_path == o[0].training_path
    assert np.array_equal(g[1], o[1]), f"{int((g[1] != o[1]).sum())} labels differ"
    for i, (x, y) in enumerate(zip(g[2] + g[3], o[2] + o[3])):
        assert np.array_equal(x, y), f"array {i} (centroid tables, then counts) differs in {int((x != y).sum())} places"

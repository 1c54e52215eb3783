"""The planned (no host round trips) and stepwise training paths must agree, including when the plan's
prediction "no random reassignment" is wrong and the run falls back (few modes, many centroids)."""
import os

import numpy as np
import pytest

from conftest import RectilinearBinMapper
from msm_we_amd import synthetic
from msm_we_amd.data import InMemorySource
from msm_we_amd.model import modelWE


def _run(h, cfg, n_clusters, planned, weights=False, random_state=7, outliers=False, shift=False):
    lat = synthetic.latent_centres(cfg)
    iters = {i: synthetic.make_iteration(cfg, i, lat) for i in range(1, cfg.n_iters + 1)}
    if outliers:
        # one very tight mode plus a few far outliers per stratum: k-means++ seeds the outliers, whose
        # counts stay ~1 while the mode's centroid collects hundreds -> count < 0.01*max -> reassignment
        rng = np.random.default_rng(99)
        for d in iters.values():
            for key in ("Xp", "Xc"):
                n = len(d[key])
                X = (rng.normal(size=(n, cfg.D)) * 0.01).astype(np.float32)
                far = rng.uniform(size=n) < 0.03
                X[far] = (rng.normal(size=(int(far.sum()), cfg.D)) * 100).astype(np.float32)
                d[key] = X
    if shift:
        # 20 well-separated modes in the first two iterations, a single mode afterwards: one centroid's count keeps
        # growing until the others fall under 1 % of it -> a reassignment many steps after the first one
        rng = np.random.default_rng(98)
        centres = rng.normal(size=(20, cfg.D)) * 50
        for it, d in iters.items():
            for key in ("Xp", "Xc"):
                n = len(d[key])
                which = rng.integers(20, size=n) if it <= 2 else np.zeros(n, dtype=int)
                d[key] = (centres[which] + rng.normal(size=(n, cfg.D)) * 0.01).astype(np.float32)
    m = modelWE(h=h)
    m.set_iteration_source(InMemorySource(iters))
    basis, target = synthetic.basis_target_bounds(cfg.S)
    m.initialize(["x"], {"coords": None, "nAtoms": cfg.D, "coord_ndim": 1}, "t", basis_pcoord_bounds=basis,
                 target_pcoord_bounds=target, tau=1.0, use_weights_in_clustering=weights)
    m.get_iterations()
    m.dimReduce()
    m.speculative_training = planned
    bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
    m.cluster_coordinates(n_clusters, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm,
                          random_state=random_state)
    cents = [cm.cluster_centers_.copy() for cm in m.clusters.cluster_models if cm.fitted]
    counts = [cm._counts.copy() for cm in m.clusters.cluster_models if cm.fitted]
    return m, np.concatenate(m.dtrajs), cents, counts


def _compare(h, cfg, n_clusters, expect_path, **kw):
    a, la, ca, na = _run(h, cfg, n_clusters, True, **kw)
    b, lb, cb, nb = _run(h, cfg, n_clusters, False, **kw)
    assert a.training_path == expect_path and b.training_path == "stepwise"
    assert np.array_equal(la, lb)
    for x, y in zip(ca + na, cb + nb):
        assert np.array_equal(x, y)
    # bookkeeping the next partial_fit would read
    for ma, mb in zip(a.clusters.cluster_models, b.clusters.cluster_models):
        if ma.fitted:
            assert ma._n_since_last_reassign == mb._n_since_last_reassign and ma.n_steps_ == mb.n_steps_


def test_planned_equals_stepwise_when_prediction_holds(oracle_h):
    _compare(oracle_h, synthetic.WEConfig("t1", 3, 30, 300, 10, 3, 12), 12, "planned")
    _compare(oracle_h, synthetic.WEConfig("t1w", 4, 30, 300, 10, 2, 12), 12, "planned", weights=True)


def test_first_step_reassignment_stays_on_the_planned_path(oracle_h):
    # a tight mode plus far outliers: the outlier seeds end the FIRST step with ~1 member and are re-seeded at once
    # (host RNG); that step is taken with the host in the loop and the rest of the pass stays fused
    _compare(oracle_h, synthetic.WEConfig("t2", 5, 12, 2500, 6, 2, 2), 20, "planned", outliers=True)


def test_late_reassignment_is_settled_and_the_plan_resumes(oracle_h):
    # a reassignment many steps in (distribution shift): the prediction "no count is low" fails, the stratum's chain
    # stops there, the host re-seeds with the model's RandomState, and from then on the permutation heads travel with
    # the plan and the device re-seeds by itself -- bit-identical to the stepwise pass
    _compare(oracle_h, synthetic.WEConfig("t2s", 5, 16, 2500, 6, 2, 2), 20, "planned", shift=True)
    _compare(oracle_h, synthetic.WEConfig("t2w", 5, 16, 2500, 6, 2, 2), 20, "planned", shift=True, weights=True)


def test_misprediction_falls_back_to_identical_result(oracle_h):
    # shared global RandomState (random_state=None -> np.random singleton, sklearn semantics): a first step that
    # draws must fall back too, the draws of different strata interleave in call order
    np.random.seed(123)
    a = _run(oracle_h, synthetic.WEConfig("t3", 6, 10, 2500, 6, 2, 2), 20, True, random_state=None, outliers=True)
    np.random.seed(123)
    b = _run(oracle_h, synthetic.WEConfig("t3", 6, 10, 2500, 6, 2, 2), 20, False, random_state=None, outliers=True)
    assert a[0].training_path == "stepwise" and np.array_equal(a[1], b[1])


@pytest.mark.gpu
def test_planned_paths_on_gpu_match_oracle(hip_h, oracle_h):
    for cfg, k, path in ((synthetic.WEConfig("t1", 3, 30, 300, 10, 3, 12), 12, "planned"),
                         (synthetic.WEConfig("t2", 5, 12, 2500, 6, 2, 2), 40, "planned+outliers"),
                         (synthetic.WEConfig("t2s", 5, 16, 2500, 6, 2, 2), 20, "planned+shift"),
                         (synthetic.WEConfig("t4", 8, 12, 3000, 30, 4, 60), 60, "planned")):
        kw = dict(outliers=path.endswith("outliers"), shift=path.endswith("shift"))
        path = path.split("+")[0]
        g = _run(hip_h, cfg, k, True, **kw)
        o = _run(oracle_h, cfg, k, True, **kw)
        assert g[0].training_path == o[0].training_path == path
        assert np.array_equal(g[1], o[1])
        for x, y in zip(g[2] + g[3], o[2] + o[3]):
            assert np.array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("rt", ["2", "4"])
def test_row_tile_variants_of_the_e_step_match_oracle(hip_h, oracle_h, rt, monkeypatch):
    """2 and 4 row tiles per E-step workgroup (chosen by operand size in production, forced here)"""
    monkeypatch.setenv("HAMSM_STEP_RT", rt)
    cfg = synthetic.WEConfig("t5", 8, 5, 12000, 30, 4, 60)
    g = _run(hip_h, cfg, 60, True)
    o = _run(oracle_h, cfg, 60, True)
    assert g[0].training_path == o[0].training_path == "planned"
    assert np.array_equal(g[1], o[1])
    for x, y in zip(g[2] + g[3], o[2] + o[3]):
        assert np.array_equal(x, y)


@pytest.mark.gpu
def test_many_strata_take_the_descriptor_table_path(hip_h, oracle_h):
    """10 strata: more steps per launch group than travel in the kernel arguments (PACK_NY = 4 per queue), so the step
    kernels read their descriptors from the table in memory and run on one queue"""
    cfg = synthetic.WEConfig("t10", 12, 8, 6000, 12, 10, 9)
    g = _run(hip_h, cfg, 9, True)
    o = _run(oracle_h, cfg, 9, True)
    assert g[0].training_path == o[0].training_path == "planned"
    assert np.array_equal(g[1], o[1])
    for x, y in zip(g[2] + g[3], o[2] + o[3]):
        assert np.array_equal(x, y)
    # and with the low-count rule firing (outliers): device-side re-seeding through the table path
    g = _run(hip_h, cfg, 9, True, outliers=True)
    o = _run(oracle_h, cfg, 9, True, outliers=True)
    assert np.array_equal(g[1], o[1])
    for x, y in zip(g[2] + g[3], o[2] + o[3]):
        assert np.array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("HAMSM_FUZZ_SEEDS", "24"))))  # (a longer soak: HAMSM_FUZZ_SEEDS=300)
def test_random_configurations_hip_equals_oracle(hip_h, oracle_h, seed):
    """seeded fuzz over the shape space (strata, clusters, features, batch sizes, weights, skewed data): the whole
    training pass and the discretization on the HIP library equal the oracle's bit for bit"""
    rng = np.random.default_rng(1000 + seed)
    S = int(rng.integers(1, 7))
    K = int(rng.integers(2, 40))
    D = int(rng.choice([1, 2, 3, 5, 8, 13, 17, 30, 33, 48, 64, 70]))
    segs = int(rng.integers(max(40, 3 * K * S), 6 * K * S + 400))
    n_it = int(rng.integers(4, 14))
    cfg = synthetic.WEConfig(f"fz{seed}", 500 + seed, n_it, segs, D, S, max(2, K // 2))
    kw = dict(weights=bool(rng.integers(2)), outliers=bool(rng.integers(3) == 0), shift=False)
    if not kw["outliers"] and rng.integers(3) == 0 and K <= 20:
        kw["shift"] = True
    try:
        o = _run(oracle_h, cfg, K, True, **kw)
    except (ValueError, AssertionError) as e:  # configurations the reference rejects too (n_samples < n_clusters, ...)
        with pytest.raises(type(e)):
            _run(hip_h, cfg, K, True, **kw)
        return
    g = _run(hip_h, cfg, K, True, **kw)
    assert g[0].training_path == o[0].training_path
    assert np.array_equal(g[1], o[1]), (cfg, K, kw)
    for x, y in zip(g[2] + g[3], o[2] + o[3]):
        assert np.array_equal(x, y), (cfg, K, kw)


@pytest.mark.gpu
@pytest.mark.parametrize("weights", [False, True])
def test_huge_clusters_span_word_blocks_and_strip_passes(hip_h, oracle_h, weights):
    """2 centroids per stratum on 4 500-row batches: ~2 250 members each -- more than one 4096-row block of bitmap
    words per centroid and several 768-member passes of the M-step's LDS strip"""
    cfg = synthetic.WEConfig("big", 21, 6, 9000, 8, 2, 2)
    g = _run(hip_h, cfg, 2, True, weights=weights)
    o = _run(oracle_h, cfg, 2, True, weights=weights)
    assert np.array_equal(g[1], o[1])
    for x, y in zip(g[2] + g[3], o[2] + o[3]):
        assert np.array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("stage_kb,rt,outliers", [("150", "1", False), ("0", "", False), ("150", "2", True), ("150", "4", False),
                                                  ("150", "", True)])
def test_wide_feature_training_matches_oracle(hip_lib, oracle_h, stage_kb, rt, outliers, monkeypatch):
    """more than 128 features per frame (the generic kernels, run-time k-step count): planned training through the model,
    with the E-step's rows staged in LDS (1, 2, 4 row tiles per workgroup, or the library's choice) and read straight from
    memory, and with reassignments (outliers) on the way"""
    from msm_we_amd import _abi

    monkeypatch.setenv("HAMSM_GENERIC_STAGE_KB", stage_kb)  # read when the context is created
    if rt:
        monkeypatch.setenv("HAMSM_GENERIC_RT", rt)
    h = _abi.Hamsm(lib=hip_lib, device=0)
    try:
        cfg = synthetic.WEConfig("w5", 8, 5, 3000, 200, 4, 30)
        g = _run(h, cfg, 30, True, outliers=outliers)
        o = _run(oracle_h, cfg, 30, True, outliers=outliers)
        assert g[0].training_path == o[0].training_path
        assert np.array_equal(g[1], o[1])
        for x, y in zip(g[2] + g[3], o[2] + o[3]):
            assert np.array_equal(x, y)
        s = _run(h, cfg, 30, False, outliers=outliers)  # stepwise on the device: same result
        assert np.array_equal(g[1], s[1])
        for x, y in zip(g[2] + g[3], s[2] + s[3]):
            assert np.array_equal(x, y)
    finally:
        h.close()


def test_a_held_cluster_model_keeps_its_random_stream(oracle_h):
    """sklearn semantics: a model's RandomState is its own for life.  A caller who keeps a StratumKMeans of a build that the
    model has since replaced still owns a live stream that CONTINUES (round 3 recycled such objects by reference counts: dropped)."""
    cfg = synthetic.WEConfig("rskeep", 21, 12, 160, 10, 3, 8)
    a, la, ca, na = _run(oracle_h, cfg, 8, True)
    kept = a.clusters.cluster_models[0]
    rs = kept._random_state
    twin = np.random.RandomState()
    twin.set_state(rs.get_state())
    bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
    for _ in range(2):   # the model replaces its clusters object twice
        a.cluster_coordinates(8, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=7)
    assert kept._random_state is rs
    assert np.array_equal(rs.randint(0, 1000, 20), twin.randint(0, 1000, 20))   # untouched by the later builds
    assert all(cm._random_state is not rs for cm in a.clusters.cluster_models)


@pytest.mark.parametrize("case", ["c1p_d10", "sparse_d10", "gap_d10", "c1p_d10_weighted"])
def test_row_list_sizes_and_starts_are_known_without_a_read_back(oracle_h, monkeypatch, case):
    """round 5: the planned pass enqueues the row-list kernel without reading its counts / starts back -- they ARE the prepass
    histogram pushed through the calls' remaps and the running sums over the strata.  `HAMSM_CHECK_ROWS=1` takes the read-back
    and asserts both equalities for every call (single- and multi-iteration calls, a never-visited bin, weights); the model
    is the golden's either way."""
    from conftest import build_model, load_golden

    g, meta = load_golden(case)
    out = []
    for check in ("1", "0"):
        monkeypatch.setenv("HAMSM_CHECK_ROWS", check)
        m, _ = build_model(oracle_h, g, meta, seed_forcing=False)
        assert m.training_path == "planned"
        out.append(np.concatenate(m.dtrajs))
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], g["dtrajs_post_cluster"])

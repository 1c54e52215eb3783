"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle cannot
finish these sizes in seconds, so it checks bounded random samples and the properties do the rest):

C2  1 M frames x 10 features, 200 clusters, 1 stratum   (assign only)
C3 10 M frames x 30 features, 4 strata x 250 clusters   (train + assign + flux + stats)
"""
import numpy as np
import pytest

from conftest import RectilinearBinMapper
from msm_we_amd import synthetic
from msm_we_amd.data import DeviceStore
from msm_we_amd.model import modelWE

pytestmark = pytest.mark.gpu


def _shard(cfg, n_iters):
    lat = synthetic.latent_centres(cfg)
    n = n_iters * cfg.segs_per_iter
    out = dict(Xp=np.empty((n, cfg.D), np.float32), Xc=np.empty((n, cfg.D), np.float32), p0=np.empty((n, 1)),
               p1=np.empty((n, 1)), w=np.empty(n))
    for k in range(n_iters):
        d = synthetic.make_iteration(cfg, k + 1, lat)
        a, b = k * cfg.segs_per_iter, (k + 1) * cfg.segs_per_iter
        for key in out:
            out[key][a:b] = d[key]
    out["offsets"] = np.arange(n_iters + 1, dtype=np.int64) * cfg.segs_per_iter
    return out


def _model(h, cfg, d):
    store = DeviceStore.from_host(h, d["Xp"], d["Xc"], d["p0"], d["p1"], d["w"], d["offsets"])
    m = modelWE(h=h)
    m.pcoord_ndim = 1
    basis, target = synthetic.basis_target_bounds(cfg.S)
    m.basis_pcoord_bounds, m.target_pcoord_bounds, m.tau = basis, target, 1.0
    m.attach_store(store)
    m.maxIter = len(d["offsets"])  # every resident iteration is complete
    m.numSegments = np.diff(d["offsets"]).astype(float)
    return m, store


def _check_assign_sample(oracle_h, h, cfg, X, p, labels, rng, n_sample=4000):
    """the oracle on a random sample of the full-size launch: labels bit-identical"""
    idx = np.sort(rng.choice(len(X), n_sample, replace=False))
    oracle_h.set_geometry(cfg.D, 1, cfg.S)
    oracle_h.set_mapper_rectilinear([synthetic.strata_boundaries(cfg.S)])
    basis, target = synthetic.basis_target_bounds(cfg.S)
    oracle_h.set_bounds(basis, target)
    for s in range(cfg.S):
        Cm, counts, csq = h.get_centroids(s)
        oracle_h.set_centroids(s, Cm, csq=csq)
    ref = oracle_h.predict_host(X[idx], p[idx])
    assert np.array_equal(ref, labels[idx])


def test_c2_assign_full_size(hip_h, oracle_h):
    cfg = synthetic.CONFIGS["c2"]  # 1000 x 1000 x 10, 1 stratum x 200
    d = _shard(cfg, cfg.n_iters)
    h = hip_h
    h.set_geometry(cfg.D, 1, cfg.S)
    h.set_mapper_rectilinear([synthetic.strata_boundaries(cfg.S)])
    basis, target = synthetic.basis_target_bounds(cfg.S)
    h.set_bounds(basis, target)
    rng = np.random.default_rng(0)
    # centroids "from a 20 k-row partial_fit" (SURVEY §8d): seed + one step on the first 20 iterations
    from msm_we_amd.stratified_clustering import StratifiedClusters

    cl = StratifiedClusters(RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)]), None, cfg.K_s, h=h, random_state=2)
    cl.cluster_models[0].partial_fit(d["Xc"][:20000])
    cl._activate()
    n = len(d["Xc"])
    d_X, d_p, d_l = h.to_device(d["Xc"]), h.to_device(d["p1"]), h.alloc(n, np.int32)
    h.assign(d_X, 0, cfg.D, d_p, n, d_l)
    lab = d_l.to_host()
    K = cfg.K_s
    tgt = d["p1"][:, 0] > cfg.S - 0.05
    bas = d["p1"][:, 0] < 0.05
    assert np.all(lab[tgt] == K + 1) and np.all(lab[bas & ~tgt] == K) and lab[~(tgt | bas)].max() < K
    _check_assign_sample(oracle_h, h, cfg, d["Xc"], d["p1"], lab, rng)
    # permutation equivariance and chunk independence (the kernel buckets frames per 512-frame chunk)
    perm = rng.permutation(n)
    h.h2d(d_X, d["Xc"][perm])
    h.h2d(d_p, d["p1"][perm])
    h.assign(d_X, 0, cfg.D, d_p, n, d_l)
    assert np.array_equal(d_l.to_host(), lab[perm])
    off = 123457
    h.assign(d_X.offset(off * cfg.D), 0, cfg.D, d_p.offset(off), 300001, d_l)
    assert np.array_equal(d_l.to_host()[:300001], lab[perm][off: off + 300001])
    # every centroid is its own nearest centroid (distance 0 to itself; ties -> first index)
    Cm = h.get_centroids(0)[0]
    assert np.array_equal(h.predict_host(Cm, np.full((K, 1), 0.5)), np.arange(K))


@pytest.fixture(scope="module")
def c3_run(hip_lib):
    from msm_we_amd import _abi

    cfg = synthetic.CONFIGS["c3"]  # 1000 x 10000 x 30, 4 strata x 250
    d = _shard(cfg, cfg.n_iters)
    h = _abi.Hamsm(lib=hip_lib)
    m, store = _model(h, cfg, d)
    bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
    m.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=1337)
    yield cfg, d, h, m, store, bm
    h.close()


def test_c3_train_assign_full_size(c3_run, oracle_h):
    cfg, d, h, m, store, bm = c3_run
    assert m.training_path == "planned" and m.n_clusters == 1000
    n = store.N  # maxIter = n_iters + 1: iterations 1..maxIter-1 (all resident ones) are trained on and discretized
    child = np.concatenate(m.dtrajs)
    parent = np.concatenate([p[:, 0] for p in m.pair_dtrajs])
    assert len(child) == n
    rng = np.random.default_rng(1)
    _check_assign_sample(oracle_h, h, cfg, d["Xc"][:n], d["p1"][:n], child.astype(np.int32), rng)
    _check_assign_sample(oracle_h, h, cfg, d["Xp"][:n], d["p0"][:n], parent.astype(np.int32), rng)
    # strata never mix: a frame of stratum s gets a label in [250 s, 250 s + 250)
    free = (d["p1"][:n, 0] >= 0.05) & (d["p1"][:n, 0] <= cfg.S - 0.05)
    s_of = np.minimum(np.floor(d["p1"][:n, 0].astype(np.float32)), cfg.S - 1).astype(int)
    assert np.array_equal(child[free] // cfg.K_s, s_of[free])
    # training bookkeeping: counts sum to the rows each stratum's model saw (sample weights are 1)
    for s, cm in enumerate(m.clusters.cluster_models):
        p0f = d["p0"][:n, 0].astype(np.float32)  # the mapper compares in float32 (westpa), basis/target in float64
        seen = ((np.floor(p0f) == s) | ((s == cfg.S - 1) & (p0f >= cfg.S))) \
            & (d["p0"][:n, 0] >= 0.05) & (d["p0"][:n, 0] <= cfg.S - 0.05)
        assert cm._counts.sum() == seen.sum()
        assert cm.n_steps_ == cfg.n_iters
    # determinism: a second full run gives bit-identical centroids and labels
    m2, _ = _model(h, cfg, d)
    m2._store = store
    m2.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=1337)
    for a, b in zip(m.clusters.cluster_models, m2.clusters.cluster_models):
        assert np.array_equal(a.cluster_centers_, b.cluster_centers_)
    assert np.array_equal(np.concatenate(m2.dtrajs), child)


def test_c3_full_size_training_equals_oracle(c3_run, oracle_lib):
    """the whole 999-call training pass of BASELINE config 3 (10 M frames, 4 strata x 250, k-means++ on the
    first call, every low-count re-seeding on the way) on the HIP library against the SAME pass on the oracle (one core,
    about half a minute): every stratum's centroids, counts, ||c||^2 and bookkeeping bit-identical.  Reference semantics:
    `_clustering.py:664-716` driving sklearn's `partial_fit` (`_kmeans.py:2203-2301`)."""
    from msm_we_amd import _abi

    cfg, d, h, m, store, bm = c3_run
    ho = _abi.Hamsm(lib=oracle_lib)
    try:
        mo, _ = _model(ho, cfg, d)
        mo.launch_ray_discretization = lambda progress_bar=None: None   # 2 x 10 M rows x 250 x 30 on a scalar core: not this test
        mo.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=1337)
        assert mo.training_path == m.training_path == "planned"
        assert [mo.clusters.we_remap[b] for b in range(cfg.S)] == [m.clusters.we_remap[b] for b in range(cfg.S)]
        for s, (ca, cb) in enumerate(zip(m.clusters.cluster_models, mo.clusters.cluster_models)):
            assert ca.cluster_centers_.shape == (cfg.K_s, cfg.D)
            assert np.array_equal(ca.cluster_centers_, cb.cluster_centers_), s
            assert np.array_equal(ca._counts, cb._counts), s
            assert ca.n_steps_ == cb.n_steps_ == cfg.n_iters
            assert ca._n_since_last_reassign == cb._n_since_last_reassign
            assert np.array_equal(h.get_centroids(s)[2], ho.get_centroids(s)[2]), s   # ||c||^2 as the step kernels keep it
    finally:
        ho.close()


def test_c3_filter_and_exact_kernel_agree_on_every_frame(c3_run, hip_lib, monkeypatch):
    """all 2 x 10 M labels of the default path (f32 ranking + exact recheck of the undecided) equal those of the exact f64
    kernel alone (HAMSM_ASSIGN_FILTER=0), and only a sliver of the frames needed the recheck"""
    import ctypes

    from msm_we_amd import _abi

    cfg, d, h, m, store, bm = c3_run
    n = store.N
    monkeypatch.setenv("HAMSM_ASSIGN_FILTER", "0")
    hx = _abi.Hamsm(lib=hip_lib)
    try:
        hx.set_geometry(cfg.D, 1, cfg.S)
        hx.set_mapper_rectilinear([synthetic.strata_boundaries(cfg.S)])
        basis, target = synthetic.basis_target_bounds(cfg.S)
        hx.set_bounds(basis, target)
        for s in range(cfg.S):
            Cm, counts, csq = h.get_centroids(s)
            hx.set_centroids(s, Cm, csq=csq)
        fn = h.lib.hamsm_debug_filter_undecided
        fn.restype = ctypes.c_longlong
        fn.argtypes = [ctypes.c_void_p]
        d_a, d_b = h.alloc(n, np.int32), hx.alloc(n, np.int32)
        for X, p in ((d["Xc"], d["p1"]), (d["Xp"], d["p0"])):
            d_X, d_p = h.to_device(X[:n]), h.to_device(p[:n])
            h.assign(d_X, 0, cfg.D, d_p, n, d_a)
            und = fn(h.ctx)
            dx_X, dx_p = hx.to_device(X[:n]), hx.to_device(p[:n])
            hx.assign(dx_X, 0, cfg.D, dx_p, n, d_b)
            assert np.array_equal(d_a.to_host(), d_b.to_host())
            assert 0 <= und < n // 1000, und
            for b in (d_X, d_p, dx_X, dx_p):
                b.free()
    finally:
        hx.close()


def test_c3_flux_and_stats_full_size(c3_run):
    cfg, d, h, m, store, bm = c3_run
    K = m.n_clusters
    n = store.N
    child = np.concatenate(m.dtrajs)
    parent = np.concatenate([p[:, 0] for p in m.pair_dtrajs])
    m.flux_mode = 1
    m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter)
    F_det = m.fluxMatrixRaw.copy()
    m.flux_mode = 0
    m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter)
    F_at = m.fluxMatrixRaw.copy()
    np.testing.assert_allclose(F_at, F_det, rtol=1e-11, atol=1e-300)
    # every iteration's weights sum to 1 and the matrix is the mean over iterations 2..maxIter-1
    assert abs(F_det.sum() - 1.0) < 1e-12
    # independent host accumulation (np.add.at, same overrides: _fluxmatrix.py:135-137)
    lo, hi = cfg.segs_per_iter, n  # iterations 2 .. maxIter-1
    a, b = parent[lo:hi].copy(), child[lo:hi].copy()
    p0, p1 = d["p0"][lo:hi, 0], d["p1"][lo:hi, 0]
    b[p1 > cfg.S - 0.05] = K + 1
    a[p0 < 0.05] = K
    b[p1 < 0.05] = K
    ref = np.zeros((K + 2) * (K + 2))
    np.add.at(ref, a * (K + 2) + b, d["w"][lo:hi])
    ref = ref.reshape(K + 2, K + 2) / (m.maxIter - 2)
    np.testing.assert_allclose(F_det, ref, rtol=1e-11, atol=1e-18)
    assert np.array_equal(F_det == 0, ref == 0)
    # determinism of the sorted kernel
    m.flux_mode = 1
    m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter)
    assert np.array_equal(m.fluxMatrixRaw, F_det)
    # cluster statistics: counts partition the frames, min <= mean <= max
    order = m.get_cluster_centers()
    cnt = m.cluster_pcoord_counts
    assert cnt.sum() == np.sum(child < K)
    c, mm = m.targetRMSD_centers[:, 0], m.targetRMSD_minmax[:, 0, :]
    ok = ~np.isnan(c)
    assert np.all(mm[ok, 0] <= c[ok] + 1e-12) and np.all(c[ok] <= mm[ok, 1] + 1e-12)
    assert np.all(np.diff(c[ok]) >= 0)  # sorted by mean pcoord
    href = np.bincount(child[child < K], weights=d["p1"][:n, 0][child < K], minlength=K) / np.maximum(
        np.bincount(child[child < K], minlength=K), 1)
    np.testing.assert_allclose(np.sort(href[np.bincount(child[child < K], minlength=K) > 0]), c[ok], rtol=1e-11)


def test_c5_shaped_plan_with_reseeding_equals_stepwise(hip_lib):
    """C5-shaped strata (more centroids than a 1024-row seeding subsample resolves: duplicate seeds, low counts):
    the fused plan -- first steps with the host in the loop, permutation heads shipped with the plan, re-seeding on the
    device, stops settled on the host -- must end bit-identical to the pass that takes every step with the host."""
    from msm_we_amd import _abi

    cfg = synthetic.WEConfig("c5m", 5, 60, 20000, 64, 4, 600)
    d = _shard(cfg, cfg.n_iters)
    h = _abi.Hamsm(lib=hip_lib)
    try:
        bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
        out = []
        for planned in (True, False):
            m, store = _model(h, cfg, d)
            m.speculative_training = planned
            m.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=11)
            out.append((m.training_path, [cm.cluster_centers_.copy() for cm in m.clusters.cluster_models],
                        [cm._counts.copy() for cm in m.clusters.cluster_models], np.concatenate(m.dtrajs),
                        [cm._n_since_last_reassign for cm in m.clusters.cluster_models]))
        assert out[0][0] == "planned" and out[1][0] == "stepwise"
        for a, b in zip(out[0][1] + out[0][2], out[1][1] + out[1][2]):
            assert np.array_equal(a, b)
        assert np.array_equal(out[0][3], out[1][3]) and out[0][4] == out[1][4]
        # the scenario is the one intended: some centroid was re-seeded (counts are no longer the member totals)
        p0f = d["p0"][:, 0].astype(np.float32)
        seen0 = ((np.floor(p0f) == 0) & (d["p0"][:, 0] >= 0.05)).sum()
        assert out[0][2][0].sum() != seen0
    finally:
        h.close()


# ---- BASELINE config 5's exact shape: D = 64, 4 strata x 1250 clusters, float32 frames, 50 000 segments per iteration ---------
# (VERDICT r1 "parity gap 1": this regime -- f32 ranking pass with grouped LDS sweeps, the 2-row-tile E-step, k-means++ with
# L = 9 candidates on a 3072-row subsample -- had only been compared with itself.)
C5_TRAIN = synthetic.WEConfig("c5t", 5, 4, 50000, 64, 4, 1250)      # training: 4 iterations = 12 500 rows per stratum and call
C5_ASSIGN = synthetic.WEConfig("c5t", 5, 24, 50000, 64, 4, 1250)    # assignment: 1.2 M segments (same generator stream)


@pytest.fixture(scope="module")
def c5_trained(hip_lib):
    """the fused (planned) training pass at C5's shape on the HIP library + the same pass on the oracle"""
    import ctypes

    from conftest import _build_oracle
    from msm_we_amd import _abi

    cfg = C5_TRAIN
    d = _shard(cfg, cfg.n_iters)
    bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
    res = {}
    hs = {"hip": _abi.Hamsm(lib=hip_lib), "oracle": _abi.Hamsm(lib=ctypes.CDLL(_build_oracle()))}
    for name, h in hs.items():
        m, store = _model(h, cfg, d)
        if name == "oracle":
            m.launch_ray_discretization = lambda progress_bar=None: None   # 2 x 200 k rows x 1250 x 64 on a scalar core: skipped
        m.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=5)
        res[name] = m
    yield cfg, d, hs, res, bm
    for h in hs.values():
        h.close()


def test_c5_shape_training_equals_oracle(c5_trained):
    """k-means++ (K = 1250, L = 9, 3072-row subsample) + 4 fused minibatch steps per stratum on 12 500-row batches:
    centroids, counts and bookkeeping bit-identical to the oracle's"""
    cfg, d, hs, res, bm = c5_trained
    a, b = res["hip"], res["oracle"]
    assert a.training_path == b.training_path == "planned"
    for s, (ca, cb) in enumerate(zip(a.clusters.cluster_models, b.clusters.cluster_models)):
        assert ca.cluster_centers_.shape == (cfg.K_s, cfg.D)
        assert np.array_equal(ca.cluster_centers_, cb.cluster_centers_), s
        assert np.array_equal(ca._counts, cb._counts), s
        assert ca._n_since_last_reassign == cb._n_since_last_reassign and ca.n_steps_ == cb.n_steps_ == cfg.n_iters
    # the discretization of the HIP run (2 x 200 k rows), oracle-checked on samples
    n = a._store.N
    rng = np.random.default_rng(15)
    orc = hs["oracle"]
    _check_assign_sample(orc, hs["hip"], cfg, d["Xc"][:n], d["p1"][:n], np.concatenate(a.dtrajs).astype(np.int32), rng, 3000)
    _check_assign_sample(orc, hs["hip"], cfg, d["Xp"][:n], d["p0"][:n],
                         np.concatenate([p[:, 0] for p in a.pair_dtrajs]).astype(np.int32), rng, 3000)


@pytest.mark.parametrize("variant", ["default", "small_lds_groups", "large_lds_groups", "bf16_full_form", "exact_only"])
def test_c5_shape_assign_million_rows_equals_oracle(c5_trained, hip_lib, monkeypatch, variant):
    """hamsm_assign over 1.2 M rows at D = 64, 4 x 1250, the centroid fragments swept through the LDS in groups: the light bf16 form
    (default at this width), the f32 form with 40 KB and 150 KB groups, the full bf16 x 3 form, and the exact f64 kernel alone:
    all labels equal each other's, and the oracle's on 4000 random rows"""
    from msm_we_amd import _abi

    cfg, d_small, hs, res, bm = c5_trained
    env = {"default": {},   # (the light bf16 form, fragments swept through LDS in groups)
           "small_lds_groups": {"HAMSM_FILTER_LDS_KB": "40", "HAMSM_FILTER_BF16": "0"},    # f32 form
           "large_lds_groups": {"HAMSM_FILTER_LDS_KB": "150", "HAMSM_FILTER_BF16": "0"},   # f32 form
           "bf16_full_form": {"HAMSM_FILTER_BF16": "2", "HAMSM_FILTER_LITE": "0"},
           "exact_only": {"HAMSM_ASSIGN_FILTER": "0"}}[variant]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    big = _shard(C5_ASSIGN, C5_ASSIGN.n_iters)
    n = len(big["w"])
    assert n >= 1_000_000
    h = _abi.Hamsm(lib=hip_lib)
    try:
        h.set_geometry(cfg.D, 1, cfg.S)
        h.set_mapper_rectilinear([synthetic.strata_boundaries(cfg.S)])
        basis, target = synthetic.basis_target_bounds(cfg.S)
        h.set_bounds(basis, target)
        for s in range(cfg.S):
            Cm, counts, csq = hs["hip"].get_centroids(s)
            h.set_centroids(s, Cm, csq=csq)
        d_X, d_p, d_l = h.to_device(big["Xc"]), h.to_device(big["p1"]), h.alloc(n, np.int32)
        h.assign(d_X, 0, cfg.D, d_p, n, d_l)
        lab = d_l.to_host()
        _check_assign_sample(hs["oracle"], h, cfg, big["Xc"], big["p1"], lab, np.random.default_rng(25), 4000)
        K = cfg.S * cfg.K_s
        tgt = big["p1"][:, 0] > cfg.S - 0.05
        bas = big["p1"][:, 0] < 0.05
        assert np.all(lab[tgt] == K + 1) and np.all(lab[bas & ~tgt] == K) and lab[~(tgt | bas)].max() < K
        ref = c5_trained[3].setdefault("_assign_ref", lab)     # the first variant's labels: every later variant must equal them
        assert np.array_equal(lab, ref)
    finally:
        h.close()


def test_wide_features_train_assign_flux_at_size(hip_lib, oracle_h):
    """D = 816 (the notebooks' flattened distance matrix, no dimensionality reduction): 40 iterations x 4000 segments through the
    generic kernels -- the oracle on a sample of both discretization passes, strata never mix, the counts add up, a second run
    is bit-identical, the flux matrix equals np.add.at on the labels, planned == stepwise training."""
    from msm_we_amd import _abi

    cfg = synthetic.WEConfig("wide816", 9, 40, 4000, 816, 4, 100)
    d = _shard(cfg, cfg.n_iters)
    h = _abi.Hamsm(lib=hip_lib)
    try:
        m, store = _model(h, cfg, d)
        bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
        m.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=5)
        assert m.n_clusters == 400
        n = store.N
        child = np.concatenate(m.dtrajs)
        parent = np.concatenate([p[:, 0] for p in m.pair_dtrajs])
        rng = np.random.default_rng(2)
        _check_assign_sample(oracle_h, h, cfg, d["Xc"][:n], d["p1"][:n], child.astype(np.int32), rng, n_sample=1500)
        _check_assign_sample(oracle_h, h, cfg, d["Xp"][:n], d["p0"][:n], parent.astype(np.int32), rng, n_sample=1500)
        free = (d["p1"][:n, 0] >= 0.05) & (d["p1"][:n, 0] <= cfg.S - 0.05)
        s_of = np.minimum(np.floor(d["p1"][:n, 0].astype(np.float32)), cfg.S - 1).astype(int)
        assert np.array_equal(child[free] // cfg.K_s, s_of[free])
        # (counts do not add up to the rows seen here: sparse clusters are re-seeded on the way, _kmeans.py:1643-1673)
        assert all(cm.n_steps_ == cfg.n_iters for cm in m.clusters.cluster_models)
        m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter)
        K = m.n_clusters
        lo = cfg.segs_per_iter  # iterations 2 .. maxIter-1 (quirk Q6), overrides of _fluxmatrix.py:135-137
        a, b = parent[lo:n].copy(), child[lo:n].copy()
        b[d["p1"][lo:n, 0] > cfg.S - 0.05] = K + 1
        a[d["p0"][lo:n, 0] < 0.05] = K
        b[d["p1"][lo:n, 0] < 0.05] = K
        F = np.zeros((K + 2) * (K + 2))
        np.add.at(F, a * (K + 2) + b, d["w"][lo:n])
        F = F.reshape(K + 2, K + 2) / (m.maxIter - 2)
        np.testing.assert_allclose(m.fluxMatrixRaw, F, rtol=1e-11, atol=1e-18)
        cents = [cm.cluster_centers_.copy() for cm in m.clusters.cluster_models]
        for planned in (True, False):
            m2, _ = _model(h, cfg, d)
            m2._store = store
            m2.speculative_training = planned
            m2.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=5)
            assert m2.training_path == ("planned" if planned else "stepwise") or planned
            for a, b in zip(cents, m2.clusters.cluster_models):
                assert np.array_equal(a, b.cluster_centers_)
            assert np.array_equal(np.concatenate(m2.dtrajs), child)
        # the first 6 iterations through the oracle as well: centroids, counts and labels bit for bit
        d6 = {k: (v[: 6 * cfg.segs_per_iter] if k != "offsets" else v[:7]) for k, v in d.items()}
        res = []
        for hh in (oracle_h, h):
            m6, _ = _model(hh, cfg, d6)
            m6.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=5)
            res.append(([cm.cluster_centers_.copy() for cm in m6.clusters.cluster_models], [cm._counts.copy() for cm in m6.clusters.cluster_models],
                        np.concatenate(m6.dtrajs)))
        for x, y in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
            assert np.array_equal(x, y)
        assert np.array_equal(res[0][2], res[1][2])
    finally:
        h.close()


# ---- BASELINE config 5 AT ITS STATED SIZE on one GPU: 1000 x 50 000 segments x 64 features, 4 x 1250 clusters ---------------
# (VERDICT r3 missing #3.  25.6 GB of frames: staged in slabs, never whole on the host; the oracle checks samples of
# regenerated iterations; the properties -- strata never mix, flux against an independent accumulation, determinism, the
# ~5000-state solve's fixed-point equations -- cover the rest.)
@pytest.fixture(scope="module")
def c5_full(hip_lib):
    from msm_we_amd import _abi

    cfg = synthetic.CONFIGS["c5"]
    h = _abi.Hamsm(lib=hip_lib)
    store, offs, host = synthetic.stage_streamed(h, cfg, 1, cfg.n_iters, keep=("p0", "p1", "w"))
    m = modelWE(h=h)
    m.pcoord_ndim = 1
    basis, target = synthetic.basis_target_bounds(cfg.S)
    m.basis_pcoord_bounds, m.target_pcoord_bounds, m.tau = basis, target, 1.0
    m.attach_store(store)
    m.maxIter = len(offs)
    m.numSegments = np.diff(offs).astype(float)
    bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
    m.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=1337,
                          store_validation_model=True)
    yield cfg, h, m, store, host, bm
    h.close()


def test_c5_full_size_train_assign(c5_full, oracle_h):
    cfg, h, m, store, host, bm = c5_full
    n = store.N
    assert n == 50_000_000 and m.n_clusters == 5000 and m.training_path == "planned"
    child = np.concatenate(m.dtrajs)
    parent = np.concatenate([p[:, 0] for p in m.pair_dtrajs])
    assert len(child) == n
    p0, p1 = host["p0"][:, 0], host["p1"][:, 0]
    K = m.n_clusters
    # basis / target short-circuits and strata never mix (stratified_clustering.py:159-196)
    for lab, p in ((child, p1), (parent, p0)):
        tgt, bas = p > cfg.S - 0.05, p < 0.05
        assert np.all(lab[tgt] == K + 1) and np.all(lab[bas & ~tgt] == K)
        free = ~(tgt | bas)
        s_of = np.minimum(np.floor(p.astype(np.float32)), cfg.S - 1).astype(np.int64)
        assert np.array_equal(lab[free] // cfg.K_s, s_of[free])
    for cm in m.clusters.cluster_models:
        assert cm.cluster_centers_.shape == (cfg.K_s, cfg.D) and cm.n_steps_ == cfg.n_iters
    # the oracle on samples of regenerated iterations, both sides (the frames themselves were never kept on the host)
    rng = np.random.default_rng(55)
    lat = synthetic.latent_centres(cfg)
    oracle_h.set_geometry(cfg.D, 1, cfg.S)
    oracle_h.set_mapper_rectilinear([synthetic.strata_boundaries(cfg.S)])
    basis, target = synthetic.basis_target_bounds(cfg.S)
    oracle_h.set_bounds(basis, target)
    for s in range(cfg.S):
        Cm, counts, csq = h.get_centroids(s)
        oracle_h.set_centroids(s, Cm, csq=csq)
    for it in rng.choice(np.arange(1, cfg.n_iters + 1), 6, replace=False):
        d = synthetic.make_iteration(cfg, int(it), lat)
        a = (int(it) - 1) * cfg.segs_per_iter
        assert np.array_equal(d["p1"][:, 0], p1[a: a + cfg.segs_per_iter])
        idx = np.sort(rng.choice(cfg.segs_per_iter, 400, replace=False))
        assert np.array_equal(oracle_h.predict_host(d["Xc"][idx], d["p1"][idx]), child[a + idx].astype(np.int32))
        assert np.array_equal(oracle_h.predict_host(d["Xp"][idx], d["p0"][idx]), parent[a + idx].astype(np.int32))
    # determinism: a second full pass on the resident frames gives bit-identical centroids and labels
    cents = [cm.cluster_centers_.copy() for cm in m.clusters.cluster_models]
    m2 = modelWE(h=h)
    m2.pcoord_ndim = 1
    m2.basis_pcoord_bounds, m2.target_pcoord_bounds, m2.tau = basis, target, 1.0
    m2.attach_store(store)
    m2.maxIter, m2.numSegments = m.maxIter, m.numSegments
    m2.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=1337)
    for a, b in zip(cents, m2.clusters.cluster_models):
        assert np.array_equal(a, b.cluster_centers_)
    assert np.array_equal(np.concatenate(m2.dtrajs), child)


def test_c5_full_size_flux_and_solve(c5_full):
    cfg, h, m, store, host, bm = c5_full
    n, K = store.N, m.n_clusters
    child = np.concatenate(m.dtrajs)
    parent = np.concatenate([p[:, 0] for p in m.pair_dtrajs])
    m.flux_mode = 1
    m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter)
    F = m.fluxMatrixRaw.copy()
    assert F.shape == (K + 2, K + 2) and abs(F.sum() - 1.0) < 1e-11
    lo = cfg.segs_per_iter  # iterations 2 .. maxIter-1 (quirk Q6), overrides of _fluxmatrix.py:135-137
    a, b = parent[lo:n].copy(), child[lo:n].copy()
    p0, p1 = host["p0"][lo:n, 0], host["p1"][lo:n, 0]
    b[p1 > cfg.S - 0.05] = K + 1
    a[p0 < 0.05] = K
    b[p1 < 0.05] = K
    ref = np.bincount(a * (K + 2) + b, weights=host["w"][lo:n], minlength=(K + 2) * (K + 2)).reshape(K + 2, K + 2) / (m.maxIter - 2)
    np.testing.assert_allclose(F, ref, rtol=1e-10, atol=1e-18)
    assert np.array_equal(F == 0, ref == 0)
    m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter)
    assert np.array_equal(m.fluxMatrixRaw, F)   # the sorted build is deterministic
    # clean -> T (CSR on the device) -> the ~5000-state steady state on the host: fixed-point equations of _analysis.py:97-191
    m.organize_fluxMatrix(use_ray=True)
    m.get_Tmatrix()
    m.get_steady_state()
    m.get_steady_state_target_flux()
    ns = m.Tmatrix.shape[0]
    assert 2000 < ns <= K + 2 and list(m.indBasis) == [ns - 2] and list(m.indTargets) == [ns - 1]
    T = m.Tmatrix_csr
    np.testing.assert_allclose(np.asarray(T.sum(axis=1)).ravel(), 1.0, rtol=1e-12)
    pi = m.pSS
    assert abs(pi.sum() - 1.0) < 1e-10 and pi.min() >= 0
    np.testing.assert_allclose(T.T @ pi, pi, rtol=1e-6, atol=1e-12)
    nt = [i for i in range(ns) if i != ns - 1]
    J = float(pi[nt] @ np.asarray(T[nt, ns - 1].todense()).ravel()) / m.tau
    assert abs(m.JtargetSS - J) <= 1e-9 * abs(J)

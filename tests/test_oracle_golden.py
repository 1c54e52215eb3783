"""Pin the CPU oracle (and the host logic of msm_we_amd, run on top of it) against golden vectors
produced by the REAL reference (tests/golden/make_golden.py): jdrusso/msm_we + scikit-learn 1.7.2.

Bars: labels / indices identical; centroids and counts bit-identical; flux, T bit-identical
(tolerance stated where it is not zero); pSS, J within 1e-6 rel (north star).
"""
import numpy as np
import pytest

from conftest import (GOLDEN_CASES, build_model, check_block_validation, check_cluster_structures, check_committor, expected_forced,
                      load_golden)


@pytest.mark.parametrize("planned", [True, False], ids=["planned", "stepwise"])
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_pipeline_matches_reference(oracle_h, case, planned):
    """planned = the whole training pass enqueued at once (prediction: no random reassignment, verified
    afterwards); stepwise = one partial_fit at a time with the host in the loop.  Same results."""
    g, meta = load_golden(case)
    m, st = build_model(oracle_h, g, meta, planned=planned)
    # k-means++ seeds: the reference's, index for index -- except where its argmin over candidate potentials is an EXACT
    # mathematical tie broken by BLAS rounding: such a seeding is teacher-forced only after `tie_witness` has
    # shown the tie in exact arithmetic, and how many there are is pinned per case
    assert st["forced"] == expected_forced(case) and all(rel == 0.0 for _, _, rel in st["witness"])
    for mine, ref in st["chosen"]:
        first_diff = np.nonzero(mine != ref)[0]
        if len(first_diff):
            assert np.array_equal(mine[: first_diff[0]], ref[: first_diff[0]])
    assert m.maxIter == meta["maxIter"]
    assert [m.clusters.we_remap[b] for b in range(meta["S"])] == list(g["remap_post_cluster"])
    assert m.n_clusters == int(g["n_clusters_post_cluster"])
    for b, cm in enumerate(m.clusters.cluster_models):
        assert cm.fitted == bool(g["fitted_post_cluster"][b])
        if cm.fitted:
            assert np.array_equal(cm.cluster_centers_, g[f"centers_post_cluster_{b}"])
            assert np.array_equal(cm._counts, g[f"counts_post_cluster_{b}"])
    assert np.array_equal(np.concatenate(m.dtrajs), g["dtrajs_post_cluster"])
    assert np.array_equal([len(d) for d in m.dtrajs], g["dtraj_sizes"])
    assert np.array_equal(np.concatenate(m.pair_dtrajs), g["pairs_post_cluster"])
    assert sorted(m.clusters.target_bins) == list(g["target_bins_post_cluster"])
    assert sorted(m.clusters.basis_bins) == list(g["basis_bins_post_cluster"])

    m.get_fluxMatrix(0, first_iter=meta.get("first_iter", 1), last_iter=m.maxIter, use_ray=True)
    assert np.array_equal(m.fluxMatrixRaw, g["fluxMatrixRaw"])

    if "organize_error" in g.files:
        # where the reference itself stops (two pcoord dimensions: `update_sorted_cluster_centers` cannot place the bin centres,
        # _clustering.py:1606) this build stops with the same exception
        with pytest.raises(Exception) as ei:
            m.organize_fluxMatrix(use_ray=True)
        assert type(ei.value).__name__ == str(g["organize_error"]).split(":")[0]
        return
    m.organize_fluxMatrix(use_ray=True)
    assert m.n_clusters == int(g["n_clusters_clean"])
    assert [m.clusters.we_remap[b] for b in range(meta["S"])] == list(g["remap_clean"])
    for b, cm in enumerate(m.clusters.cluster_models):
        if f"centers_clean_{b}" in g.files:
            assert np.array_equal(cm.cluster_centers_, g[f"centers_clean_{b}"])
    assert np.array_equal(np.concatenate(m.dtrajs), g["dtrajs_clean"])
    assert np.array_equal(np.concatenate(m.pair_dtrajs), g["pairs_clean"])
    assert np.array_equal(m.fluxMatrix, g["fluxMatrix"])
    assert np.array_equal(m.fluxMatrixRaw, g["fluxMatrixRaw"])  # restored pre-clean matrix (_clustering.py:1123)
    # cluster pcoord statistics: mean to 1e-12 rel (reference uses pairwise np.nanmean), min/max exact
    np.testing.assert_allclose(m.targetRMSD_centers[:-2, 0], g["targetRMSD_centers"][:-2, 0], rtol=1e-12)
    assert np.array_equal(m.targetRMSD_minmax[:-2], g["targetRMSD_minmax"][:-2])
    np.testing.assert_allclose(m.all_centers, g["all_centers"], rtol=1e-12)
    assert np.array_equal(m.indBasis, g["indBasis"]) and np.array_equal(m.indTargets, g["indTargets"])

    m.get_Tmatrix()
    assert np.array_equal(m.Tmatrix, g["Tmatrix"])
    m.get_steady_state()
    np.testing.assert_allclose(m.pSS, g["pSS"], rtol=1e-6, atol=1e-300)
    m.get_steady_state_target_flux()
    assert abs(m.JtargetSS - float(g["JtargetSS"])) <= 1e-6 * float(g["JtargetSS"])
    check_committor(m, g)  # a13
    check_cluster_structures(m, g)  # f4
    check_block_validation(m, g)  # a14, against the reference's own groups


def test_every_partial_fit_matches_reference(oracle_h):
    """Centres and counts after EVERY partial_fit of the reference run (76 calls), teacher-forced
    only at the one tie-flipped seeding of c1p_d30."""
    from msm_we_amd.stratified_clustering import StratumKMeans

    for case in ("c1p_d10", "c1p_d30", "c1p_d10_weighted"):
        g, meta = load_golden(case)
        log = []
        orig = StratumKMeans.partial_fit_device

        def rec(self, *a, **k):
            out = orig(self, *a, **k)
            Cm, counts, _ = self._h.get_centroids(self.stratum)
            log.append((self.stratum, Cm, counts, self._n_since_last_reassign))
            return out

        StratumKMeans.partial_fit_device = rec
        try:
            build_model(oracle_h, g, meta, planned=False)
        finally:
            StratumKMeans.partial_fit_device = orig
        assert len(log) == len(g["fit_bin"])
        for i, (s, Cm, counts, n_since) in enumerate(log):
            assert s == g["fit_bin"][i]
            assert np.array_equal(Cm, g["fit_centers"][i]), (case, i)
            assert np.array_equal(counts, g["fit_counts"][i]), (case, i)
            assert n_since == g["fit_n_since"][i]


def test_unforced_seeding_is_deterministic(oracle_h):
    """Without teacher forcing the oracle still runs end to end and is reproducible."""
    g, meta = load_golden("c1p_d30")
    a, _ = build_model(oracle_h, g, meta, seed_forcing=False)
    la = np.concatenate(a.dtrajs)
    b, _ = build_model(oracle_h, g, meta, seed_forcing=False)
    assert np.array_equal(la, np.concatenate(b.dtrajs))


def test_npz_directory_source_round_trip(oracle_h, tmp_path):
    """the h5py-free ingest path: a golden case exported as per-iteration .npz files gives the same model"""
    from conftest import golden_iterations
    from msm_we_amd.data import InMemorySource, NpzDirectorySource

    g, meta = load_golden("c1p_d10")
    src = NpzDirectorySource.export(InMemorySource(golden_iterations(g)), tmp_path / "iters")
    assert src.iterations() == sorted(golden_iterations(g))
    m, _ = build_model(oracle_h, g, meta, source=src)
    assert np.array_equal(np.concatenate(m.dtrajs), g["dtrajs_post_cluster"])


def test_pca_pipeline_matches_reference(oracle_h):
    """dim_reduce_method='pca' (golden c1p_d30_pca): the IncrementalPCA fitted by the reference's two-pass scheme is
    reproduced bit for bit (same estimator, same batches); its transform runs through `hamsm_project`, whose plain
    fma chains differ from BLAS by ~1 ulp -- so centroids carry that ulp-level difference, while every discrete
    result (seeds, labels, cleaning, sparsity pattern) must still be the reference's."""
    g, meta = load_golden("c1p_d30_pca")
    m, st = build_model(oracle_h, g, meta)
    assert m.ndim == int(g["pca_ndim"]) and m.coordinates.n_samples_seen_ == int(g["pca_n_samples_seen"])
    assert np.array_equal(m.coordinates.components_, g["pca_components"]) and np.array_equal(m.coordinates.mean_, g["pca_mean"])
    assert m._store.D == int(g["pca_ndim"])
    # the seedings are compared by their chosen INDICES (hamsm_kmeans_seed_chosen): on the projected features, which differ from
    # the reference's by an ulp, this build's k-means++ picks the reference's rows -- none is teacher-forced
    assert st["forced"] == 0 and len(st["chosen"]) == 2 and all(np.array_equal(a, b) for a, b in st["chosen"])
    assert [m.clusters.we_remap[b] for b in range(meta["S"])] == list(g["remap_post_cluster"])
    for b, cm in enumerate(m.clusters.cluster_models):
        if cm.fitted:
            np.testing.assert_allclose(cm.cluster_centers_, g[f"centers_post_cluster_{b}"], rtol=1e-10, atol=1e-12)
            assert np.array_equal(cm._counts, g[f"counts_post_cluster_{b}"])
    assert np.array_equal(np.concatenate(m.dtrajs), g["dtrajs_post_cluster"])
    assert np.array_equal(np.concatenate(m.pair_dtrajs), g["pairs_post_cluster"])
    m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter, use_ray=True)
    assert np.array_equal(m.fluxMatrixRaw, g["fluxMatrixRaw"])
    m.organize_fluxMatrix(use_ray=True)
    assert m.n_clusters == int(g["n_clusters_clean"]) and np.array_equal(np.concatenate(m.dtrajs), g["dtrajs_clean"])
    assert np.array_equal(m.fluxMatrix, g["fluxMatrix"])
    m.get_Tmatrix()
    assert np.array_equal(m.Tmatrix, g["Tmatrix"])
    m.get_steady_state()
    np.testing.assert_allclose(m.pSS, g["pSS"], rtol=1e-6, atol=1e-300)
    m.get_steady_state_target_flux()
    assert abs(m.JtargetSS - float(g["JtargetSS"])) <= 1e-6 * float(g["JtargetSS"])


def check_c3_shaped(h, g, meta, n_states=1000):
    """BASELINE config 3's geometry (D = 30, 4 strata x 250 clusters, 10 000 segments per iteration, the bench generator's
    seed) over 12 iterations against the REAL reference's outputs for exactly these inputs (golden c3s; inputs regenerated
    from the seed): every label of both passes, every centroid table, both flux matrices, T's input and the block-validation
    groups identical; steady state / target flux within 1e-6 rel.  No k-means++ seeding is teacher-forced here."""
    m, st = build_model(h, g, meta, seed_forcing=False)
    assert m.n_clusters == int(g["n_clusters_post_cluster"]) == n_states
    for b, cm in enumerate(m.clusters.cluster_models):
        assert np.array_equal(cm.cluster_centers_, g[f"centers_post_cluster_{b}"])
        assert np.array_equal(cm._counts, g[f"counts_post_cluster_{b}"])
    assert np.array_equal(np.concatenate(m.dtrajs), g["dtrajs_post_cluster"])
    assert np.array_equal(np.concatenate(m.pair_dtrajs), g["pairs_post_cluster"])
    m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter, use_ray=True)
    assert np.array_equal(m.fluxMatrixRaw, g["fluxMatrixRaw"])
    m.organize_fluxMatrix(use_ray=True)
    assert m.n_clusters == int(g["n_clusters_clean"])
    assert np.array_equal(np.concatenate(m.dtrajs), g["dtrajs_clean"])
    assert np.array_equal(np.concatenate(m.pair_dtrajs), g["pairs_clean"])
    assert np.array_equal(m.fluxMatrix, g["fluxMatrix"])
    m.get_Tmatrix()
    m.get_steady_state()
    np.testing.assert_allclose(m.pSS, g["pSS"], rtol=1e-6, atol=1e-300)
    m.get_steady_state_target_flux()
    assert abs(m.JtargetSS - float(g["JtargetSS"])) <= 1e-6 * float(g["JtargetSS"])
    check_committor(m, g)
    check_cluster_structures(m, g)
    check_block_validation(m, g)


def test_c3_shaped_slice_matches_reference(oracle_h):
    g, meta = load_golden("c3s")
    check_c3_shaped(oracle_h, g, meta)


def test_more_than_128_features_match_reference(oracle_h):
    """golden wide_d160: 160 features per frame, no dimensionality reduction (the notebooks' flattened distance-matrix
    featurizers are wider still) -- the same end-to-end comparison against the real reference's outputs"""
    g, meta = load_golden("wide_d160")
    assert meta["D"] == 160
    check_c3_shaped(oracle_h, g, meta, n_states=50)


def plain_euclid(coordvec, centers):
    """the same distance function WITHOUT the device marker: the mapper's own assign runs on the host"""
    d = np.asarray(centers, dtype=np.float64) - np.asarray(coordvec, dtype=np.float64)
    return np.sqrt(np.sum(d * d, axis=1))


def check_voronoi_external(h):
    """VoronoiBinMapper whose dfunc the library does not know: bins from the mapper's own `assign` on the host
    (hamsm_set_mapper_external / hamsm_set_frame_bins) -- same outputs as the reference run of golden voronoi_d10, and
    `clusters.predict` (B3) goes through the same path"""
    from conftest import golden_iterations

    g, meta = load_golden("voronoi_d10")
    m, st = build_model(h, g, meta, mapper_dfunc=plain_euclid)
    assert m._mapper_external and st["forced"] == 0
    assert [m.clusters.we_remap[b] for b in range(meta["S"])] == list(g["remap_post_cluster"])
    assert np.array_equal(np.concatenate(m.dtrajs), g["dtrajs_post_cluster"])
    assert np.array_equal(np.concatenate(m.pair_dtrajs), g["pairs_post_cluster"])
    m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter, use_ray=True)
    assert np.array_equal(m.fluxMatrixRaw, g["fluxMatrixRaw"])
    it = 9
    m.load_iter_data(it)
    assert np.array_equal(m.clusters.predict(golden_iterations(g)[it]["Xc"]), m.dtrajs[it - 1])
    m.organize_fluxMatrix(use_ray=True)
    assert np.array_equal(np.concatenate(m.dtrajs), g["dtrajs_clean"]) and np.array_equal(m.fluxMatrix, g["fluxMatrix"])


def test_voronoi_mapper_with_user_dfunc_runs_on_the_host(oracle_h):
    check_voronoi_external(oracle_h)


def test_unsupported_mapper_is_refused_like_the_reference(oracle_h):
    """`_clustering.py:598-609`: functional mappers may re-bin on the fly; only the two static ones are accepted"""
    g, meta = load_golden("c1p_d10")

    class FuncBinMapper:
        nbins = 2

    from msm_we_amd.data import InMemorySource
    from msm_we_amd.model import modelWE
    from conftest import golden_iterations

    m = modelWE(h=oracle_h)
    m.set_iteration_source(InMemorySource(golden_iterations(g)))
    m.initialize(["g.h5"], {"coords": None, "nAtoms": meta["D"], "coord_ndim": 1}, "g", basis_pcoord_bounds=meta["basis"],
                 target_pcoord_bounds=meta["target"], tau=1.0, pcoord_ndim=1)
    m.get_iterations()
    m.get_coordSet(m.maxIter)
    m.dimReduce()
    with pytest.raises(Exception, match="supported mappers"):
        m.cluster_coordinates(25, streaming=True, use_ray=True, stratified=True, user_bin_mapper=FuncBinMapper())


def test_nan_coordinates_fail_where_the_reference_fails(oracle_h):
    """a NaN frame outside basis / target reaches sklearn's input validation in the reference (ValueError at discretization);
    NaN coordinates in the FIRST clustering call are fatal there (AssertionError, `_clustering.py:693-701`)"""
    from conftest import golden_iterations
    from msm_we_amd.data import InMemorySource

    g, meta = load_golden("c1p_d10")
    its = {k: {a: np.array(b, copy=True) for a, b in v.items()} for k, v in golden_iterations(g).items()}
    mid = np.nonzero((its[9]["p1"][:, 0] > 0.5) & (its[9]["p1"][:, 0] < 1.5))[0][0]
    its[9]["Xc"][mid, 3] = np.nan
    with pytest.raises(ValueError, match="NaN"):
        build_model(oracle_h, g, meta, source=InMemorySource(its))
    its = {k: {a: np.array(b, copy=True) for a, b in v.items()} for k, v in golden_iterations(g).items()}
    its[1]["Xc"][0, 0] = np.nan
    with pytest.raises(AssertionError):
        build_model(oracle_h, g, meta, source=InMemorySource(its))


def test_random_state_from_an_integer_seed_is_numpys():
    """`_check_random_state(int)` builds its RandomState without the SeedSequence detour: same type, same state, same draws as
    `np.random.RandomState(seed)` (sklearn.utils.check_random_state)"""
    import pickle

    from msm_we_amd.stratified_clustering import _check_random_state

    for seed in (0, 1, 1337, 2 ** 31, 2 ** 32 - 1, np.int64(42), np.uint32(7)):
        a, b = np.random.RandomState(seed), _check_random_state(seed)
        assert type(b) is np.random.RandomState
        sa, sb = a.get_state(), b.get_state()
        assert sa[0] == sb[0] and np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
        assert np.array_equal(a.randint(0, 10 ** 6, 50), b.randint(0, 10 ** 6, 50))
        assert a.choice(100, p=np.full(100, 0.01)) == b.choice(100, p=np.full(100, 0.01))
        assert np.array_equal(a.uniform(size=9), b.uniform(size=9)) and a.standard_normal() == b.standard_normal()
        c = pickle.loads(pickle.dumps(b))           # (models are pickled with their stream)
        assert np.array_equal(c.permutation(20), b.permutation(20))
    with pytest.raises((ValueError, TypeError, OverflowError)):
        _check_random_state(-1)                     # as np.random.RandomState(-1)

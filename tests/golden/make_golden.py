#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (msm_we @ /root/reference) on small
synthetic WE-shaped inputs.  Run in the build container only:

    OMP_NUM_THREADS=1 python tests/golden/make_golden.py [case ...]

Outputs ``tests/golden/<case>.npz`` (inputs + every observable of the hot path).  The tests only
read those files.  Versions of numpy/scipy/sklearn are recorded in each file (parity target is
scikit-learn 1.7.2 as installed; SURVEY.md §7 hard part 6).
"""
from __future__ import annotations

import json
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

from msm_we_amd import synthetic  # noqa: E402
import ref_harness  # noqa: E402

# case -> (generator config, n_iters override, per-case options)
CASES = {
    # C1' of SURVEY §8d: 40 iters x 128 segs, 2 strata x 25 clusters, seed 1337
    "c1p_d10": dict(cfg="c1p_d10"),
    "c1p_d30": dict(cfg="c1p_d30"),
    # sample weights in clustering (`_clustering.py:853-854,898-899`)
    "c1p_d10_weighted": dict(cfg="c1p_d10", use_weights=True),
    # a stratum that is never visited -> we_remap / find_nearest_bin (`_clustering.py:721-731`), quirk Q3
    "gap_d10": dict(cfg=synthetic.WEConfig("gap_d10", 77, 30, 160, 10, 4, 12), drop_stratum=2),
    # multi-iteration accumulation: too few segments per iteration to fill a stratum at once (`:794-886`)
    "sparse_d10": dict(cfg=synthetic.WEConfig("sparse_d10", 5, 60, 24, 10, 2, 20)),
    # dimensionality reduction: IncrementalPCA fitted by the reference's two-pass scheme (`_dimensionality.py:139-241`),
    # every frame transformed before clustering (`reduceCoordinates`)
    "c1p_d30_pca": dict(cfg="c1p_d30", dim_reduce="pca"),
    # VoronoiBinMapper (SUPPORTED_MAPPERS, _clustering.py:598-609): Euclidean distance function, four centres of which the
    # last is never visited -> `find_nearest_bin`'s Voronoi branch (:1362-1365)
    "voronoi_d10": dict(cfg="c1p_d10", voronoi=[[0.3], [0.9], [1.6], [5.0]]),
    # NaN-coordinate segments (`_data.py:303-313, 549-553`): the child frames of the target-bound segments of iterations 12 and
    # 30 and the parent frames of the basis-bound segments of iteration 20 are NaN -- weight 0 in the flux matrix, their
    # iterations lost for training (AssertionError swallowed, `_clustering.py:693-705`), discretized by pcoord alone
    "nan_d10": dict(cfg="c1p_d10", nan_child_in_target=[12, 30], nan_parent_in_basis=[20]),
    # two progress coordinates, a 3 x 2 rectilinear grid (row-major bin index, `find_nearest_bin`'s meshgrid of bin centres,
    # `_clustering.py:1367-1381`), strict 2-D basis / target boxes (`msm_we.py:462-527`); one grid cell is never visited
    "grid2d_d10": dict(cfg="c1p_d10", grid2d=True),
    # training / flux restricted to later iterations: `first_cluster_iter` (`_clustering.py:640-651`) and `first_iter` of
    # get_fluxMatrix (`_fluxmatrix.py:205-215`), as HAMSMDriver's `first_analysis_iter` sets them (hamsm_driver.py:122-131)
    "late_start_d10": dict(cfg="c1p_d10", first_iter=7),
    # C3-SHAPED slice: BASELINE config 3's geometry -- D = 30, 4 strata x 250 clusters, 10 000 segments
    # per iteration, the bench generator's own seed -- over 12 iterations.  Inputs are NOT stored (the tests regenerate
    # them from the seed with msm_we_amd.synthetic, 29 MB); outputs are stored compactly (labels int16, flux as COO).
    "c3s": dict(cfg=synthetic.WEConfig("c3s", 3, 12, 10000, 30, 4, 250), compact=True, block_validation=(2, 2)),
    # more than 128 features per frame (the notebooks' flattened distance-matrix featurizers reach 816): 160 features, no
    # dimensionality reduction -- the library's generic kernels (run-time k-step count).  Inputs regenerated from the seed.
    "wide_d160": dict(cfg=synthetic.WEConfig("wide_d160", 11, 20, 128, 160, 2, 25), compact=True),
}


def build_inputs(case):
    spec = CASES[case]
    cfg = spec["cfg"] if isinstance(spec["cfg"], synthetic.WEConfig) else synthetic.CONFIGS[spec["cfg"]]
    lat = synthetic.latent_centres(cfg)
    iters = {}
    for it in range(1, cfg.n_iters + 1):
        d = synthetic.make_iteration(cfg, it, lat)
        if "drop_stratum" in spec:  # carve a hole in pcoord space: move those frames to stratum+1
            s = spec["drop_stratum"]
            for key, xkey in (("p0", "Xp"), ("p1", "Xc")):
                m = np.floor(d[key][:, 0]) == s
                d[key][m, 0] += 1.0
        if spec.get("grid2d"):
            # second progress coordinate: the sign pattern of two features, in {-1.5, -0.5, 0.5, 1.5}; the cell
            # (first coordinate in [1, 2), second >= 0) is emptied -> a never-visited bin
            for key, xkey in (("p0", "Xp"), ("p1", "Xc")):
                q = (d[xkey][:, 0] > 0).astype(np.float64) + (d[xkey][:, 1] > 0).astype(np.float64) * 0.5 - 0.75
                q = q * 2.0
                hole = (np.floor(d[key][:, 0]) == 1) & (q >= 0)
                q[hole] = -q[hole] - 0.25
                d[key] = np.concatenate([d[key], q[:, None]], axis=1)
        if it in spec.get("nan_child_in_target", ()):
            d["Xc"][d["p1"][:, 0] > cfg.S - 0.05] = np.nan
        if it in spec.get("nan_parent_in_basis", ()):
            d["Xp"][d["p0"][:, 0] < 0.05] = np.nan
        iters[it] = d
    return cfg, iters


def compact(out):
    """C3-shaped case: drop what the tests can regenerate or do not assert, shrink what remains.  Labels fit int16 (K + 2 <=
    1002), dense (K+2)^2 matrices become COO triplets (exact: float64 data), per-partial_fit centroid logs keep the final
    state of every stratum only."""
    keep = {}
    for k, v in out.items():
        if k.startswith("in_") and k != "in_offsets":
            continue   # regenerated from the seed
        if k in ("fit_centers", "pairs_post_cluster", "pairs_clean", "Tmatrix", "ucs_iter_seg", "ucs_pcoords", "ucs_weights"):
            continue   # (pairs = parent labels + dtrajs: parent labels kept below; T is a function of fluxMatrix)
        v = np.asarray(v)
        if v.ndim == 2 and v.shape[0] == v.shape[1] and v.shape[0] > 64 and v.dtype == np.float64:
            r, c = np.nonzero(v)
            keep[k + "__coo_rc"] = np.stack([r, c]).astype(np.int32)
            keep[k + "__coo_v"] = v[r, c]
            keep[k + "__coo_n"] = np.int64(v.shape[0])
            continue
        if v.dtype == np.int64 and v.ndim == 1 and v.size > 4096 and v.min() >= 0 and v.max() < 32768:
            v = v.astype(np.int16)
        keep[k] = v
    for tag in ("post_cluster", "clean"):
        keep[f"parents_{tag}"] = np.asarray(out[f"pairs_{tag}"])[:, 0].astype(np.int16)
    return keep


def run_case(case):
    spec = CASES[case]
    cfg, iters = build_inputs(case)
    msm_we = ref_harness.load_reference()
    import sklearn
    import scipy
    import sklearn.cluster._kmeans as skk
    import msm_we.stratified_clustering as sc

    seed = 1337
    log = {"fit": [], "pp": []}

    # --- recorders -----------------------------------------------------------------------
    orig_pp = skk._kmeans_plusplus

    def rec_pp(X, n_clusters, x_squared_norms, sample_weight, random_state, n_local_trials=None):
        centers, indices = orig_pp(X, n_clusters, x_squared_norms, sample_weight, random_state, n_local_trials)
        log["pp"].append(dict(n=X.shape[0], indices=np.array(indices), centers=centers.copy()))
        return centers, indices

    skk._kmeans_plusplus = rec_pp

    class RecordingKMeans(sc.mini_kmeans.__mro__[0]):
        _bin = -1

        def partial_fit(self, X, y=None, sample_weight=None):
            first = not hasattr(self, "cluster_centers_")
            npp = len(log["pp"])
            out = super().partial_fit(X, y, sample_weight=sample_weight)
            log["fit"].append(
                dict(
                    bin=self._bin,
                    n=X.shape[0],
                    first=first,
                    pp=(npp if first else -1),
                    xsum=float(np.sum(X)),
                    centers=self.cluster_centers_.copy(),
                    counts=self._counts.copy(),
                    n_since=self._n_since_last_reassign,
                )
            )
            return out

    RecordingKMeans.__name__ = "MiniBatchKMeans"
    RecordingKMeans.__qualname__ = "MiniBatchKMeans"
    base_cls = sc.mini_kmeans

    def factory(**kw):
        m = RecordingKMeans(**kw)
        m._bin = factory.count
        factory.count += 1
        return m

    factory.count = 0
    sc.mini_kmeans = factory

    # --- model ---------------------------------------------------------------------------
    Model = ref_harness.make_model_class(msm_we)
    Model._iters = iters
    Model.seg_weights = {}
    model = Model()
    basis, target = synthetic.basis_target_bounds(cfg.S)
    ndim = 1
    if spec.get("grid2d"):
        ndim = 2
        basis = np.array([[-np.inf, 0.05], [-np.inf, np.inf]])
        target = np.array([[cfg.S - 0.05, np.inf], [-2.0, 2.0]])
    first_it = int(spec.get("first_iter", 1))
    model.initialize(
        ["in-memory.h5"],
        {"coords": np.zeros((cfg.D, 1)), "nAtoms": cfg.D, "coord_ndim": 1},
        "golden",
        basis_pcoord_bounds=basis,
        target_pcoord_bounds=target,
        dim_reduce_method=spec.get("dim_reduce", "none"),
        tau=1e-9,
        pcoord_ndim=ndim,
        use_weights_in_clustering=bool(spec.get("use_weights", False)),
    )
    model.get_iterations()
    model.get_coordSet(last_iter=model.maxIter)
    model.dimReduce()
    out = {}
    if spec.get("dim_reduce") == "pca":
        out["pca_components"] = np.asarray(model.coordinates.components_, dtype=np.float64)
        out["pca_mean"] = np.asarray(model.coordinates.mean_, dtype=np.float64)
        out["pca_ndim"] = np.int64(model.ndim)
        out["pca_n_samples_seen"] = np.int64(model.coordinates.n_samples_seen_)
    bm = ref_harness.RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
    if spec.get("grid2d"):
        bm = ref_harness.RectilinearBinMapper([np.array([0, 1, 2, np.inf], dtype=np.float32), np.array([-np.inf, 0, np.inf], dtype=np.float32)])
    if "voronoi" in spec:
        from msm_we_amd.binning import euclidean_dfunc

        bm = ref_harness.VoronoiBinMapper(euclidean_dfunc, np.array(spec["voronoi"], dtype=np.float32))
    model.cluster_coordinates(
        n_clusters=cfg.K_s, streaming=True, use_ray=True, stratified=True,
        user_bin_mapper=bm, random_state=seed, store_validation_model=True,
        **({"first_cluster_iter": first_it} if first_it != 1 else {}),
    )
    out["remap_post_cluster"] = np.array([model.clusters.we_remap[b] for b in range(bm.nbins)], dtype=np.int64)
    out["n_clusters_post_cluster"] = np.int64(model.n_clusters)
    fitted = [hasattr(m, "cluster_centers_") for m in model.clusters.cluster_models]
    out["fitted_post_cluster"] = np.array(fitted)
    for b, m in enumerate(model.clusters.cluster_models):
        if fitted[b]:
            out[f"centers_post_cluster_{b}"] = m.cluster_centers_.copy()
            out[f"counts_post_cluster_{b}"] = m._counts.copy()
    out["dtrajs_post_cluster"] = np.concatenate(model.dtrajs).astype(np.int64)
    out["dtraj_sizes"] = np.array([len(d) for d in model.dtrajs], dtype=np.int64)
    out["pairs_post_cluster"] = np.concatenate([np.array(p) for p in model.pair_dtrajs]).astype(np.int64)
    out["target_bins_post_cluster"] = np.array(sorted(model.clusters.target_bins), dtype=np.int64)
    out["basis_bins_post_cluster"] = np.array(sorted(model.clusters.basis_bins), dtype=np.int64)

    model.get_fluxMatrix(0, first_iter=first_it, last_iter=model.maxIter, use_ray=True)
    out["fluxMatrixRaw"] = model.fluxMatrixRaw.copy()

    try:  # (the reference's own limits are recorded, not hidden: e.g. two pcoord dimensions fail in update_sorted_cluster_centers)
        model.organize_fluxMatrix(use_ray=True)
        out["fluxMatrix"] = model.fluxMatrix.copy()
        out["n_clusters_clean"] = np.int64(model.n_clusters)
        out["remap_clean"] = np.array([model.clusters.we_remap[b] for b in range(bm.nbins)], dtype=np.int64)
        for b, m in enumerate(model.clusters.cluster_models):
            if hasattr(m, "cluster_centers_"):
                out[f"centers_clean_{b}"] = m.cluster_centers_.copy()
        out["dtrajs_clean"] = np.concatenate(model.dtrajs).astype(np.int64)
        out["pairs_clean"] = np.concatenate([np.array(p) for p in model.pair_dtrajs]).astype(np.int64)
        out["targetRMSD_centers"] = np.asarray(model.targetRMSD_centers, dtype=np.float64)
        out["targetRMSD_minmax"] = np.asarray(model.targetRMSD_minmax, dtype=np.float64)
        out["all_centers"] = np.asarray(model.all_centers, dtype=np.float64)
        out["indBasis"] = np.asarray(model.indBasis)
        out["indTargets"] = np.asarray(model.indTargets)
        # fluxMatrixRaw of the cleaned model before the pcoord sort (recomputed: organize restores the pre-clean one)
        model.get_Tmatrix()
        out["Tmatrix"] = model.Tmatrix.copy()
        model.get_steady_state()
        out["pSS"] = np.asarray(model.pSS).copy()
        model.get_steady_state_target_flux()
        out["JtargetSS"] = np.float64(model.JtargetSS)

        # --- committor (_analysis.py:527-606) on the organised model ---------------------------------------------------------
        try:
            model.get_committor()
            out["committor"] = np.asarray(model.q, dtype=np.float64)
        except Exception as e:  # noqa: BLE001
            out["committor_error"] = np.array(f"{type(e).__name__}: {e}"[:300])

        # --- f4: update_cluster_structures (_clustering.py:1398-1526) on the organised model -----------------------------------
        try:
            model.update_cluster_structures(build_pcoord_cache=True)
            keys = list(model.cluster_structures.keys())
            out["ucs_keys"] = np.array(keys, dtype=np.int64)
            out["ucs_counts"] = np.array([len(model.cluster_structures[k]) for k in keys], dtype=np.int64)
            out["ucs_weights"] = np.concatenate([np.asarray(model.cluster_structure_weights[k], dtype=np.float64) for k in keys])
            out["ucs_coord_sums"] = np.array([np.sum(np.asarray(model.cluster_structures[k], dtype=np.float64)) for k in keys])
            out["ucs_first_coord"] = np.stack([np.asarray(model.cluster_structures[k][0], dtype=np.float64).ravel() for k in keys])
            out["ucs_iter_seg"] = np.concatenate([np.array([[a, b] for a, b, _ in model.structure_iteration_segments[k]], dtype=np.int64)
                                                  for k in keys])
            out["ucs_pcoords"] = np.concatenate([np.asarray(model.pcoord_cache[k], dtype=np.float64).reshape(len(model.pcoord_cache[k]), -1)
                                                 for k in keys])
        except Exception as e:  # noqa: BLE001
            out["ucs_error"] = np.array(f"{type(e).__name__}: {e}"[:300])

        # --- a14: do_block_validation (msm_we.py:884-1009): per group flux matrix -> clean -> T -> steady state -> J ----------
        groups, blocks = spec.get("block_validation", (2, 4))
        try:
            model.do_block_validation(cross_validation_groups=groups, cross_validation_blocks=blocks, use_ray=True)
            out["bv_ok"] = np.bool_(True)
            out["bv_groups_blocks"] = np.array([groups, blocks], dtype=np.int64)
            for g, (its_g, vm) in enumerate(zip(model.validation_iterations, model.validation_models)):
                out[f"bv_iters_{g}"] = np.array(list(its_g), dtype=np.int64)
                out[f"bv_fluxMatrixRaw_{g}"] = np.asarray(vm.fluxMatrixRaw, dtype=np.float64)
                out[f"bv_fluxMatrix_{g}"] = np.asarray(vm.fluxMatrix, dtype=np.float64)
                out[f"bv_n_clusters_{g}"] = np.int64(vm.n_clusters)
                out[f"bv_dtrajs_{g}"] = np.concatenate(vm.dtrajs).astype(np.int64)
                out[f"bv_pSS_{g}"] = np.asarray(vm.pSS, dtype=np.float64)
                out[f"bv_JtargetSS_{g}"] = np.float64(vm.JtargetSS)
                out[f"bv_remap_{g}"] = np.array([vm.clusters.we_remap[b] for b in range(bm.nbins)], dtype=np.int64)
        except Exception as e:  # noqa: BLE001 -- recorded: the tests then expect the same failure class
            out["bv_ok"] = np.bool_(False)
            out["bv_error"] = np.array(f"{type(e).__name__}: {e}"[:500])
    except Exception as e:  # noqa: BLE001
        if "fluxMatrix" in out and "pSS" in out:
            raise
        out["organize_error"] = np.array(f"{type(e).__name__}: {e}"[:300])

    # --- training log ----------------------------------------------------------------------
    out["fit_bin"] = np.array([f["bin"] for f in log["fit"]], dtype=np.int64)
    out["fit_n"] = np.array([f["n"] for f in log["fit"]], dtype=np.int64)
    out["fit_first"] = np.array([f["first"] for f in log["fit"]])
    out["fit_xsum"] = np.array([f["xsum"] for f in log["fit"]])
    out["fit_n_since"] = np.array([f["n_since"] for f in log["fit"]], dtype=np.int64)
    out["fit_centers"] = np.stack([f["centers"] for f in log["fit"]])
    out["fit_counts"] = np.stack([f["counts"] for f in log["fit"]])
    out["fit_pp"] = np.array([f["pp"] for f in log["fit"]], dtype=np.int64)
    out["pp_indices"] = np.stack([p["indices"] for p in log["pp"]]).astype(np.int64)
    out["pp_centers"] = np.stack([p["centers"] for p in log["pp"]])
    out["pp_n"] = np.array([p["n"] for p in log["pp"]], dtype=np.int64)

    # --- inputs ----------------------------------------------------------------------------
    its = sorted(iters)
    for k in ("Xp", "Xc", "p0", "p1", "w"):
        out["in_" + k] = np.concatenate([iters[i][k] for i in its])
    sizes = np.array([iters[i]["w"].shape[0] for i in its], dtype=np.int64)
    out["in_offsets"] = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    meta = dict(
        case=case, seed=seed, voronoi=spec.get("voronoi"), grid2d=bool(spec.get("grid2d")), first_iter=first_it,
        pcoord_ndim=ndim, boundaries_nd=[[float(x) for x in b] for b in bm.boundaries] if hasattr(bm, "boundaries") else None,
        gen_seed=cfg.seed, segs_per_iter=cfg.segs_per_iter, D=cfg.D,
        S=int(bm.nbins), K_s=cfg.K_s,
        n_iters=cfg.n_iters, maxIter=int(model.maxIter),
        tau=1e-9, use_weights=bool(spec.get("use_weights", False)), dim_reduce=spec.get("dim_reduce", "none"),
        boundaries=[float(x) for x in synthetic.strata_boundaries(cfg.S)],
        basis=basis.tolist(), target=target.tolist(),
        numpy=np.__version__, scipy=scipy.__version__, sklearn=sklearn.__version__,
        reference="jdrusso/msm_we 0.1.28.dev1 @ /root/reference", omp_num_threads=os.environ.get("OMP_NUM_THREADS"),
    )
    out["meta"] = np.array(json.dumps(meta))
    if spec.get("compact"):
        out = compact(out)
    sc.mini_kmeans = base_cls
    skk._kmeans_plusplus = orig_pp
    path = os.path.join(HERE, f"{case}.npz")
    np.savez_compressed(path, **out)
    print(
        f"{case}: K={out['n_clusters_post_cluster']} -> clean {out.get('n_clusters_clean', out.get('organize_error'))}, "
        f"{len(log['fit'])} partial_fits, J={float(out.get('JtargetSS', np.nan)):.6e}, file {os.path.getsize(path)/1e6:.2f} MB"
    )


if __name__ == "__main__":
    cases = sys.argv[1:] or list(CASES)
    for c in cases:
        # one process per case: the harness patches module globals
        if len(cases) > 1:
            import subprocess

            subprocess.check_call([sys.executable, os.path.abspath(__file__), c])
        else:
            run_case(c)

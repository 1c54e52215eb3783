#!/usr/bin/env python3
"""Differential fuzz pack: the REAL reference (msm_we @ /root/reference, through tests/golden/ref_harness.py) run on many small
random WE-shaped inputs -- random numbers of iterations, segments, features, strata, clusters per stratum, with and without
sample weights, with and without a stratum emptied -- and every discrete observable of the hot path recorded, including the
stage at which the reference itself gives up (too few segments to fill a bin on the first call, everything cleaned away, ...).

    OMP_NUM_THREADS=1 python tests/golden/make_fuzz.py [n_cases]      ->  tests/golden/fuzz_pack.npz

Inputs are not stored: tests/test_reference_fuzz.py regenerates them from the recorded generator parameters."""
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


def case_spec(i):
    rng = np.random.default_rng(90_000 + i)
    S = int(rng.integers(1, 6))
    spec = dict(gen_seed=int(1000 + i), n_iters=int(rng.integers(10, 36)), segs=int(rng.integers(12, 140)), D=int(rng.integers(2, 13)),
                S=S, K_s=int(rng.integers(2, 11)), use_weights=bool(rng.integers(2)), seed=int(rng.integers(0, 2 ** 31 - 1)),
                drop_stratum=int(rng.integers(0, S)) if S > 2 and rng.random() < 0.3 else -1)
    # one case in five bins with a Voronoi mapper (Euclidean distance to S random centres) instead of the unit intervals
    spec["voronoi"] = sorted(float(np.float32(x)) for x in rng.uniform(0.0, S + 0.5, size=S)) if S > 1 and rng.random() < 0.2 else None
    if i >= 64:  # cases 64 ..: frames of more than 128 features (the library's generic kernels)
        spec["D"] = int(rng.integers(129, 320))
    return spec


def case_inputs(spec):
    cfg = synthetic.WEConfig("fuzz", spec["gen_seed"], spec["n_iters"], spec["segs"], spec["D"], spec["S"], spec["K_s"])
    lat = synthetic.latent_centres(cfg)
    iters = {}
    for it in range(1, cfg.n_iters + 1):
        d = synthetic.make_iteration(cfg, it, lat)
        if spec["drop_stratum"] >= 0:
            for key in ("p0", "p1"):
                m = np.floor(d[key][:, 0]) == spec["drop_stratum"]
                d[key][m, 0] += 1.0
        iters[it] = d
    return cfg, iters


def run(i, msm_we, skk, sc):
    spec = case_spec(i)
    cfg, iters = case_inputs(spec)
    pp = []
    orig_pp = skk._kmeans_plusplus

    def rec_pp(X, n_clusters, x_squared_norms, sample_weight, random_state, n_local_trials=None):
        centers, indices = orig_pp(X, n_clusters, x_squared_norms, sample_weight, random_state, n_local_trials)
        pp.append((np.array(indices), centers.copy()))
        return centers, indices

    skk._kmeans_plusplus = rec_pp
    out = {}
    try:
        Model = ref_harness.make_model_class(msm_we)
        Model._iters = iters
        Model.seg_weights = {}
        model = Model()
        basis, target = synthetic.basis_target_bounds(cfg.S)
        model.initialize(["in-memory.h5"], {"coords": np.zeros((cfg.D, 1)), "nAtoms": cfg.D, "coord_ndim": 1}, "fuzz",
                         basis_pcoord_bounds=basis, target_pcoord_bounds=target, dim_reduce_method="none", tau=1e-9, pcoord_ndim=1,
                         use_weights_in_clustering=spec["use_weights"])
        model.get_iterations()
        model.get_coordSet(last_iter=model.maxIter)
        model.dimReduce()
        bm = ref_harness.RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
        if spec["voronoi"]:
            from msm_we_amd.binning import euclidean_dfunc

            bm = ref_harness.VoronoiBinMapper(euclidean_dfunc, np.array(spec["voronoi"], dtype=np.float32)[:, None])
        stage = "cluster"
        try:
            model.cluster_coordinates(n_clusters=cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm,
                                      random_state=spec["seed"])
            out["remap_post"] = np.array([model.clusters.we_remap[b] for b in range(bm.nbins)], dtype=np.int16)
            out["n_clusters_post"] = np.int64(model.n_clusters)
            out["fitted"] = np.array([hasattr(m, "cluster_centers_") for m in model.clusters.cluster_models])
            out["dtrajs_post"] = np.concatenate(model.dtrajs).astype(np.int16)
            out["parents_post"] = np.concatenate([np.array(p)[:, 0] for p in model.pair_dtrajs]).astype(np.int16)
            stage = "flux"
            model.get_fluxMatrix(0, first_iter=1, last_iter=model.maxIter, use_ray=True)
            out["fluxMatrixRaw"] = model.fluxMatrixRaw.copy()
            stage = "organize"
            model.organize_fluxMatrix(use_ray=True)
            out["n_clusters_clean"] = np.int64(model.n_clusters)
            out["remap_clean"] = np.array([model.clusters.we_remap[b] for b in range(bm.nbins)], dtype=np.int16)
            out["dtrajs_clean"] = np.concatenate(model.dtrajs).astype(np.int16)
            out["fluxMatrix"] = model.fluxMatrix.copy()
            stage = "solve"
            model.get_Tmatrix()
            model.get_steady_state()
            model.get_steady_state_target_flux()
            out["pSS"] = np.asarray(model.pSS, dtype=np.float64)
            out["J"] = np.float64(model.JtargetSS)
            stage = "done"
        except Exception as e:  # noqa: BLE001 -- the stage and class of the reference's own failure are part of the record
            out["error"] = np.array(f"{stage}|{type(e).__name__}|{str(e)[:160]}")
        spec["stage"] = stage
        spec["maxIter"] = int(model.maxIter)
    finally:
        skk._kmeans_plusplus = orig_pp
    if pp:
        out["pp_n"] = np.array([len(a) for a, _ in pp], dtype=np.int64)
        out["pp_indices"] = np.concatenate([a for a, _ in pp]).astype(np.int64)
        out["pp_centers"] = np.concatenate([c for _, c in pp])
    out["meta"] = np.array(json.dumps(spec))
    return out, spec


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 72
    import logging

    msm_we = ref_harness.load_reference()
    logging.getLogger().setLevel(logging.ERROR)
    logging.getLogger("msm_we").setLevel(logging.CRITICAL)
    import sklearn
    import sklearn.cluster._kmeans as skk
    import msm_we.stratified_clustering as sc

    pack = {"n_cases": np.int64(n), "versions": np.array(json.dumps(dict(numpy=np.__version__, sklearn=sklearn.__version__)))}
    stages = {}
    for i in range(n):
        out, spec = run(i, msm_we, skk, sc)
        for k, v in out.items():
            pack[f"c{i:02d}_{k}"] = v
        stages[spec["stage"]] = stages.get(spec["stage"], 0) + 1
        print(f"case {i:02d}: {spec}", flush=True)
    path = os.path.join(HERE, "fuzz_pack.npz")
    np.savez_compressed(path, **pack)
    print(f"{n} cases, stages reached: {stages}, file {os.path.getsize(path) / 1e6:.2f} MB")

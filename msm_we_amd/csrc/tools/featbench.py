"""Throughput of hamsm_featurize_dist (a1): frames x atoms float32 coordinates -> per-atom distances to a reference."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np
from msm_we_amd import _abi

h = _abi.Hamsm()
rng = np.random.default_rng(0)
for n, n_atoms, n_sel in ((1_000_000, 272, 80), (2_000_000, 80, 80), (4_000_000, 10, 10)):
    coords = rng.normal(size=(n, n_atoms * 3)).astype(np.float32)
    idx = np.sort(rng.choice(n_atoms, n_sel, replace=False)).astype(np.int32)
    ref = rng.normal(size=(n_sel, 3)).astype(np.float32)
    d_c = h.to_device(coords)
    d_X = h.alloc(n * n_sel, np.float64)
    h.featurize_dist(d_c, _abi.F32, 3 * n_atoms, n, n_atoms, idx, ref, d_X)
    h.sync()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        h.featurize_dist(d_c, _abi.F32, 3 * n_atoms, n, n_atoms, idx, ref, d_X)
    h.sync()
    dt = (time.perf_counter() - t0) / reps
    alg = n * (12 * n_sel + 8 * n_sel)          # selected atoms in, distances out
    touched = n * (12 * n_atoms + 8 * n_sel)    # whole frames in (what the lines of a sparse selection cost)
    print(f"{n} frames x {n_atoms} atoms, {n_sel} selected: {dt * 1e3:.3f} ms, {alg / dt / 1e9:.0f} GB/s algorithmic, "
          f"{touched / dt / 1e9:.0f} GB/s of touched frames", flush=True)
    d_c.free(); d_X.free()

"""throughput of hamsm_project (PCA transform) on synthetic raw features"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
from msm_we_amd import _abi
h = _abi.Hamsm()
rng = np.random.default_rng(0)
for n, Draw, d in ((2_000_000, 300, 30), (1_000_000, 900, 50), (4_000_000, 60, 10)):
    X = rng.normal(size=(n, Draw)).astype(np.float32)
    W = rng.normal(size=(d, Draw)); mu = rng.normal(size=Draw)
    dX = h.to_device(X); dY = h.alloc(n * d, np.float64)
    h.project(dX, 0, Draw, n, W, mu, None, dY); h.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        h.project(dX, 0, Draw, n, W, mu, None, dY)
    h.sync()
    dt = (time.perf_counter() - t0) / 3
    fl = 2.0 * n * Draw * d
    by = n * (Draw * 4 + d * 8)
    print(f"n={n} Draw={Draw} d={d}: {dt*1e3:.2f} ms, {fl/dt/1e12:.1f} TFLOP/s algorithmic, {by/dt/1e9:.0f} GB/s algorithmic")
    dX.free(); dY.free()

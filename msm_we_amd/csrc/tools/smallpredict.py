"""Latency of small assignment calls (the B3 use: WESTPA calls `predict` on a few hundred rows every iteration): predict_host
(host buffers in and out) and hamsm_assign on resident rows, C3's model shape.  usage: python msm_we_amd/csrc/tools/smallpredict.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
from msm_we_amd import _abi  # noqa: E402

D, S, K_s = 30, 4, 250
h = _abi.Hamsm()
h.set_geometry(D, 1, S)
b = np.arange(S + 1, dtype=np.float32)
b[-1] = np.inf
h.set_mapper_rectilinear([b])
h.set_bounds([[-np.inf, 0.05]], [[S - 0.05, np.inf]])
rng = np.random.default_rng(0)
for s in range(S):
    Cm = rng.normal(size=(K_s, D)) * 3
    h.set_centroids(s, Cm, csq=np.einsum("ij,ij->i", Cm, Cm))
for n in (1, 64, 300, 1000, 4096, 20000, 100000):
    X = (rng.normal(size=(n, D)) * 3).astype(np.float32)
    p = rng.uniform(0, S, size=(n, 1))
    h.predict_host(X, p)
    t0 = time.perf_counter()
    for _ in range(200):
        h.predict_host(X, p)
    t1 = time.perf_counter()
    d_X, d_p, d_l = h.to_device(X), h.to_device(p), h.alloc(n, np.int32)
    h.assign(d_X, _abi.F32, D, d_p, n, d_l)
    t2 = time.perf_counter()
    for _ in range(200):
        h.assign(d_X, _abi.F32, D, d_p, n, d_l)
    t3 = time.perf_counter()
    print(f"n = {n:7d}: predict_host {1e6 * (t1 - t0) / 200:8.1f} us per call, assign (resident) {1e6 * (t3 - t2) / 200:8.1f} us per call")
    for x in (d_X, d_p, d_l):
        x.free()

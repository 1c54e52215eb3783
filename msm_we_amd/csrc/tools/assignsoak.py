"""Repeat hamsm_assign on the same C3-shaped data and compare every call's labels with the first call's (the ranking pass appends
undecided rows through atomics and defers stores: a race would show as a label that changes between calls).
usage: python msm_we_amd/csrc/tools/assignsoak.py [calls] [rows]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
from msm_we_amd import _abi  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
for D, S, K, dt in ((30, 4, 250, np.float32), (64, 4, 1250, np.float32), (13, 3, 40, np.float64)):
    h = _abi.Hamsm()
    h.set_geometry(D, 1, S)
    b = np.arange(S + 1, dtype=np.float32)
    b[-1] = np.inf
    h.set_mapper_rectilinear([b])
    h.set_bounds([[-np.inf, 0.05]], [[S - 0.05, np.inf]])
    rng = np.random.default_rng(1)
    cents = [rng.normal(size=(K, D)) * 2 for _ in range(S)]
    for s in range(S):
        h.set_centroids(s, cents[s])
    p = rng.uniform(0, S, size=(n, 1))
    # frames near centroids of their stratum (most rows certain) plus a share of diffuse ones (near ties -> undecided)
    X = np.empty((n, D), dtype=dt)
    st = np.minimum(p[:, 0].astype(int), S - 1)
    for s in range(S):
        m = np.nonzero(st == s)[0]
        X[m] = (cents[s][rng.integers(K, size=len(m))] + rng.normal(size=(len(m), D)) * 0.7).astype(dt)
    far = rng.uniform(size=n) < 0.05
    X[far] = rng.normal(size=(int(far.sum()), D)).astype(dt) * 2
    dX, dp, dl = h.to_device(X), h.to_device(p), h.alloc(n, np.int32)
    xd = _abi.F32 if dt == np.float32 else _abi.F64
    h.assign(dX, xd, D, dp, n, dl)
    ref = dl.to_host().copy()
    und = h.lib.hamsm_debug_filter_undecided(h.ctx)
    bad = 0
    for c in range(calls):
        h.memset(dl, 0xFF, n * 4) if hasattr(h, "memset") else None
        h.assign(dX, xd, D, dp, n, dl)
        lab = dl.to_host()
        if not np.array_equal(lab, ref):
            bad += 1
            print(f"  call {c}: {int((lab != ref).sum())} labels differ", flush=True)
    print(f"D={D} S={S} K_s={K} {np.dtype(dt).name}: {calls} calls of {n} rows, undecided per call {und}, calls that differ: {bad}", flush=True)
    h.close()

"""Soak of the pipelined training pass against the oracle: random shapes, starts, slacks; every case must be
bit-identical (labels, centroid tables, counts).  usage: python aheadsoak.py [n_cases] [first_seed]"""
import ctypes, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from msm_we_amd import _abi, synthetic
from test_training_paths import _run
from conftest import _build_oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
h = _abi.Hamsm()
o = _abi.Hamsm(lib=ctypes.CDLL(_build_oracle()))
h.lib.hamsm_debug_set_ahead.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
h.lib.hamsm_debug_ahead_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
bad = 0
t0 = time.time()
for case in range(seed0, seed0 + n_cases):
    rng = np.random.default_rng(case)
    D = int(rng.choice([2, 3, 6, 10, 17, 30, 33, 64, 100]))
    S = int(rng.integers(1, 7))
    K = int(rng.integers(2, 41))
    n_it = int(rng.integers(16, 70))
    segs = int(rng.integers(max(200, 3 * K * S), 3000))
    weights = bool(rng.integers(2))
    kind = rng.choice(["plain", "plain", "plain", "shift", "outliers"])
    alpha = float(rng.choice([1.0, 1.5, 3.0, 8.0]))
    n0 = int(rng.choice([2, 5, 12]))
    cfg = synthetic.WEConfig(f"soak{case}", 100 + case, n_it, segs, D, S, K)
    kw = dict(weights=weights, shift=kind == "shift", outliers=kind == "outliers")
    if kind != "plain":
        cfg = synthetic.WEConfig(f"soak{case}", 100 + case, n_it, max(segs, 2500), min(D, 10), min(S, 2), 2)
        K = int(rng.choice([20, 40]))
    try:
        r = _run(o, cfg, K, True, **kw)
        h.lib.hamsm_debug_set_ahead(h.ctx, 1, n0, 16, alpha)
        out = (ctypes.c_longlong * 6)()
        h.lib.hamsm_debug_ahead_stats(h.ctx, out, 1)
        g = _run(h, cfg, K, True, **kw)
        h.lib.hamsm_debug_ahead_stats(h.ctx, out, 1)
        same = g[0].training_path == r[0].training_path and np.array_equal(g[1], r[1]) and all(
            np.array_equal(a, b) for a, b in zip(g[2] + g[3], r[2] + r[3]))
    except Exception as e:  # noqa: BLE001
        same = False
        print("case", case, "raised", repr(e)[:200], flush=True)
    bad += not same
    print(f"case {case}: D={cfg.D} S={cfg.S} K={K} it={cfg.n_iters} segs={cfg.segs_per_iter} w={weights} {kind} alpha={alpha} n0={n0} "
          f"segments={out[0]} steps_in={out[1]} failed={out[2]} classic={out[5]} -> {'ok' if same else 'MISMATCH'}", flush=True)
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)

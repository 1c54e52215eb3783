"""Look-ahead blocks of the training pass: the E-steps of several consecutive WE iterations of a stratum run in
one launch against the centroids at the block's start; labels are final only where a displacement bound proves them, the other
rows are decided exactly by the M-step, every M-step checks the bound.  The results must be those of the step-by-step path --
i.e. of the oracle -- bit for bit, whatever the block sizes, and also when blocks are given up and relaunched."""
import ctypes

import numpy as np
import pytest

from msm_we_amd import synthetic
from test_training_paths import _run

pytestmark = [pytest.mark.gpu, pytest.mark.experiments]   # csrc/experiments/: `make EXPERIMENTS=1`, `pytest -m experiments`


@pytest.fixture()
def hip_h(exp_lib):
    """these tests drive the experiments' library, not the shipped one"""
    from msm_we_amd import _abi

    h = _abi.Hamsm(lib=exp_lib, device=0)
    yield h
    h.close()


def _ahead(h, on=1, n0=2, gmax=16, alpha=1.5):
    h.lib.hamsm_debug_set_ahead.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    h.lib.hamsm_debug_set_ahead.restype = ctypes.c_int
    assert h.lib.hamsm_debug_set_ahead(h.ctx, on, n0, gmax, alpha) == 0


def _stats(h, reset=True):
    out = (ctypes.c_longlong * 6)()
    h.lib.hamsm_debug_ahead_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    h.lib.hamsm_debug_ahead_stats.restype = ctypes.c_int
    assert h.lib.hamsm_debug_ahead_stats(h.ctx, out, int(reset)) == 0
    return dict(blocks=out[0], steps_in_blocks=out[1], given_up=out[2], relaunches=out[4], steps_classic=out[5])


def _same(g, o, what=""):
    assert g[0].training_path == o[0].training_path, what
    assert np.array_equal(g[1], o[1]), f"{what}: {int((g[1] != o[1]).sum())} labels differ"
    for i, (x, y) in enumerate(zip(g[2] + g[3], o[2] + o[3])):
        assert np.array_equal(x, y), f"{what}: array {i} (centroid tables, then counts) differs in {int((x != y).sum())} places"


CASES = [
    # name, config (seed, iterations, segments, D, S, K_s), clusters, kwargs
    ("balanced", synthetic.WEConfig("la1", 3, 60, 1500, 10, 3, 12), 12, {}),
    ("c3-like", synthetic.WEConfig("la2", 8, 48, 4000, 30, 4, 60), 60, {}),
    ("weighted", synthetic.WEConfig("la3", 4, 50, 1200, 10, 2, 12), 12, dict(weights=True)),
    ("wide-rows", synthetic.WEConfig("la4", 11, 40, 1600, 100, 2, 20), 20, {}),
    ("d64", synthetic.WEConfig("la5", 12, 40, 2400, 64, 3, 40), 40, {}),
    ("many-strata", synthetic.WEConfig("la6", 13, 40, 3300, 12, 11, 8), 8, {}),
    ("more-centroids-than-modes", synthetic.WEConfig("la7", 5, 40, 2500, 6, 2, 2), 20, {}),
    ("reseeding", synthetic.WEConfig("la8", 5, 30, 2500, 6, 2, 2), 20, dict(shift=True)),
    ("outliers", synthetic.WEConfig("la9", 5, 24, 2500, 6, 2, 2), 40, dict(outliers=True)),
]


@pytest.mark.parametrize("name,cfg,k,kw", CASES, ids=[c[0] for c in CASES])
def test_lookahead_blocks_equal_the_oracle(hip_h, oracle_h, name, cfg, k, kw):
    o = _run(oracle_h, cfg, k, True, **kw)
    try:
        for gmax, alpha in ((16, 1.5), (4, 1.0), (16, 8.0)):
            _ahead(hip_h, 1, 2, gmax, alpha)
            _stats(hip_h)
            g = _run(hip_h, cfg, k, True, **kw)
            st = _stats(hip_h)
            _same(g, o, f"{name} gmax={gmax} alpha={alpha} {st}")
            assert st["blocks"] > 0 and st["steps_in_blocks"] > 0, (name, st)
        _ahead(hip_h, 0)
        g = _run(hip_h, cfg, k, True, **kw)
        assert _stats(hip_h)["blocks"] == 0
        _same(g, o, f"{name} look-ahead off")
    finally:
        _ahead(hip_h, 0, 200, 16, 3.0)


def test_lookahead_serves_most_steps_of_a_separated_run(hip_h, oracle_h):
    """well separated modes (the bench's generator): after the warm-up the pass runs in blocks, few of them given up"""
    cfg = synthetic.WEConfig("la10", 3, 160, 4000, 30, 4, 40)
    try:
        _ahead(hip_h, 1, 16, 16, 1.5)
        _stats(hip_h)
        g = _run(hip_h, cfg, 40, True)
        st = _stats(hip_h)
    finally:
        _ahead(hip_h, 0, 200, 16, 3.0)
    _same(g, _run(oracle_h, cfg, 40, True))
    total = st["steps_in_blocks"] + st["steps_classic"]
    assert st["steps_in_blocks"] > 0.5 * total, st

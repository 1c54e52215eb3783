"""(An experiment that is off by default)  A training call's E-step can run through the assignment kernels
(ranking pass under a proven bound + exact f64 kernel, `hamsm_assign_rows`).  The labels are the contract's either way, so the
pass must stay bit-identical to the oracle -- here with that path forced on shapes of every kind."""
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


def _ranked(h, mode, min_mflop=-1.0):
    h.lib.hamsm_debug_set_ranked.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
    h.lib.hamsm_debug_set_ranked.restype = ctypes.c_longlong
    return int(h.lib.hamsm_debug_set_ranked(h.ctx, mode, min_mflop))


CASES = [
    ("balanced", synthetic.WEConfig("rk1", 3, 30, 1500, 10, 3, 12), 12, {}),
    ("c3-like", synthetic.WEConfig("rk2", 8, 24, 4000, 30, 4, 60), 60, {}),
    ("weighted", synthetic.WEConfig("rk3", 4, 30, 1200, 10, 2, 12), 12, dict(weights=True)),
    ("wide-rows", synthetic.WEConfig("rk4", 11, 20, 1600, 100, 2, 20), 20, {}),
    ("d64", synthetic.WEConfig("rk5", 12, 20, 2400, 64, 3, 40), 40, {}),
    ("many-strata", synthetic.WEConfig("rk6", 13, 20, 3300, 12, 11, 8), 8, {}),
    ("more-centroids-than-modes", synthetic.WEConfig("rk7", 5, 20, 2500, 6, 2, 2), 20, {}),
    ("reseeding", synthetic.WEConfig("rk8", 5, 16, 2500, 6, 2, 2), 20, dict(shift=True)),
    ("outliers", synthetic.WEConfig("rk9", 5, 12, 2500, 6, 2, 2), 40, dict(outliers=True)),
]


@pytest.mark.parametrize("name,cfg,k,kw", CASES, ids=[c[0] for c in CASES])
def test_ranked_estep_equals_the_oracle(hip_h, oracle_h, name, cfg, k, kw):
    o = _run(oracle_h, cfg, k, True, **kw)
    try:
        _ranked(hip_h, 2)
        g = _run(hip_h, cfg, k, True, **kw)
        served = _ranked(hip_h, 0)
        assert served > 0, name
        assert g[0].training_path == o[0].training_path
        assert np.array_equal(g[1], o[1]), f"{name}: {int((g[1] != o[1]).sum())} labels differ"
        for i, (x, y) in enumerate(zip(g[2] + g[3], o[2] + o[3])):
            assert np.array_equal(x, y), f"{name}: array {i} (centroid tables, then counts) differs"
        g0 = _run(hip_h, cfg, k, True, **kw)      # ... and the same with the path off
        assert _ranked(hip_h, 0) == 0
        assert np.array_equal(g0[1], o[1])
    finally:
        _ranked(hip_h, 0)

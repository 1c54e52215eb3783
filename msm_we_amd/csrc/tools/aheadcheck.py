"""(debug) run one look-ahead test case on a library built with -DAHEAD_CHECK (csrc/_var/lib_chk.so): every open row's distance
left behind by the previous step's wave is recomputed by the consuming wave and differences are printed."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from msm_we_amd import _abi, synthetic  # noqa: E402
from test_training_paths import _run  # noqa: E402
from conftest import _build_oracle  # noqa: E402

lib = ctypes.CDLL(os.environ.get("AHEAD_LIB") or os.path.join(ROOT, "msm_we_amd", "csrc", "_var", "lib_chk.so"))
h = _abi.Hamsm(lib=lib)
o = _abi.Hamsm(lib=ctypes.CDLL(_build_oracle()))
h.lib.hamsm_debug_set_ahead.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
cfg = synthetic.WEConfig("la1", 3, 60, 1500, 10, 3, 12)
r = _run(o, cfg, 12, True)
for gmax, alpha in ((16, 1.5), (4, 1.0), (16, 8.0)):
    h.lib.hamsm_debug_set_ahead(h.ctx, 1, 2, gmax, alpha)
    g = _run(h, cfg, 12, True)
    print(gmax, alpha, "labels equal:", np.array_equal(g[1], r[1]), "centroids equal:", all(np.array_equal(a, b) for a, b in zip(g[2], r[2])),
          "counts equal:", all(np.array_equal(a, b) for a, b in zip(g[3], r[3])), flush=True)

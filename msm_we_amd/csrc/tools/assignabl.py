"""What bounds the ranking pass at C5's shape (64 features, 1250 centroids per stratum): hamsm_assign on 10 M random frames with the
shipped library and with diagnostic builds that leave out the sweep's LDS fragment reads / its ranking instructions (wrong labels,
timing only -- and only the FIRST calls of a variant count: once most frames come back undecided the library steps over to its
other forms).  Build first:  for v in NOFRAG NORANK; do tools/build_var.sh abl_$v "-DFILT_ABL_$v" assign; done; run under
tools/assignabl.sh for the kernel's own durations."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
from msm_we_amd import _abi  # noqa: E402

D, S, K, n = 64, 4, 1250, 10_000_000
rng = np.random.default_rng(1)
cents = [rng.normal(size=(K, D)) * 3 for _ in range(S)]
p = rng.uniform(0.06, S - 0.06, size=(n, 1))
X = np.empty((n, D), dtype=np.float32)   # every frame next to a centroid of its stratum: the ranking pass certifies (nearly) all
for s in range(S):
    m = np.floor(p[:, 0]).astype(int) == s
    X[m] = (cents[s][rng.integers(0, K, int(m.sum()))] + 0.05 * rng.normal(size=(int(m.sum()), D))).astype(np.float32)
for name in ("base", "abl_NOFRAG", "abl_NORANK"):
    path = _abi.library_path() if name == "base" else os.path.join(ROOT, "msm_we_amd/csrc/_var", f"lib_{name}.so")
    if not os.path.exists(path):
        print(name, "not built")
        continue
    h = _abi.Hamsm(lib=ctypes.CDLL(path))
    h.set_geometry(D, 1, S)
    b = np.arange(S + 1, dtype=np.float32)
    b[-1] = np.inf
    h.set_mapper_rectilinear([b])
    h.set_bounds([[-np.inf, 0.05]], [[S - 0.05, np.inf]])
    for s in range(S):
        h.set_centroids(s, cents[s], csq=np.einsum("ij,ij->i", cents[s], cents[s]))
    d_X, d_p, d_lab = h.to_device(X), h.to_device(p), h.alloc(n, np.int32)
    for _ in range(2):
        h.assign(d_X, _abi.F32, D, d_p, n, d_lab)
    h.timers(reset=True)
    reps = 5
    for _ in range(reps):
        h.assign(d_X, _abi.F32, D, d_p, n, d_lab)
    h.sync()
    t = h.timers()
    fn = h.lib.hamsm_debug_filter_undecided
    fn.restype = ctypes.c_longlong
    fn.argtypes = [ctypes.c_void_p]
    print(f"{name:12s} undecided {fn(h.ctx)}", end="  ")
    print(f"assign {t['assign'] / reps * 1e3:7.3f} ms per {n} rows (bucket passes {t['other'] / reps * 1e3:.3f} ms)", flush=True)
    h.close()

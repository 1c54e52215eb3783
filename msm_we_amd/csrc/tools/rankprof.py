#!/usr/bin/env python3
"""Where a workgroup of the ranked E-step (k_step_assign_ranked) spends its time: phase stamps of thread 0 of every workgroup,
summed (100 MHz ticks).  Needs the diagnostic build:  tools/build_var.sh rankprof "-DRANK_PROFILE" train
    python msm_we_amd/csrc/tools/rankprof.py [iterations] [segments] [D] [K_s]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from msm_we_amd import _abi, synthetic  # noqa: E402
from msm_we_amd.data import DeviceStore  # noqa: E402
from msm_we_amd.model import modelWE  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 40
segs = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
K_s = int(sys.argv[4]) if len(sys.argv) > 4 else 1250
lib = ctypes.CDLL(os.path.join(ROOT, "msm_we_amd", "csrc", "_var", os.environ.get("RANKPROF_LIB", "lib_rankprof.so")))
h = _abi.Hamsm(lib=lib)
cfg = synthetic.WEConfig("rp", 5, n_it, segs, D, 4, K_s)
blk = synthetic.make_block(cfg, 1, n_it)
store = DeviceStore.from_host(h, blk["Xp"], blk["Xc"], blk["p0"], blk["p1"], blk["w"], blk["offsets"])


class Mapper:
    def __init__(self, b):
        self.boundaries = [np.asarray(x, dtype=np.float32) for x in b]
        self.nbins = len(self.boundaries[0]) - 1


Mapper.__name__ = "RectilinearBinMapper"
prof = lib.hamsm_debug_rank_profile
prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["rows + pieces", "sweep (MFMA + ranking)", "merge + barrier", "per-row decision (two-candidate chains)", "exact sweeps", "labels + bitmap"]
for rep in range(2):
    m = modelWE(h=h)
    m.pcoord_ndim = 1
    basis, target = synthetic.basis_target_bounds(cfg.S)
    m.basis_pcoord_bounds, m.target_pcoord_bounds, m.tau = basis, target, 1.0
    m.attach_store(store)
    m.maxIter = n_it + 1
    m.numSegments = np.diff(blk["offsets"]).astype(float)
    m.launch_ray_discretization = lambda progress_bar=None: None
    prof(None, 1)
    h.timers(reset=True)
    m.cluster_coordinates(K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=Mapper([synthetic.strata_boundaries(4)]), random_state=5)
    out = (ctypes.c_ulonglong * 16)()
    prof(out, 0)
    n = max(1, out[8])
    print(f"pass {rep}: {n} workgroups; training steps {h.timers()['step'] * 1e3:.2f} ms")
    for i, nm in enumerate(names):
        print(f"   {nm:45s} {out[i] / n / 100:8.2f} us per workgroup")
    print(f"   longest exact-sweep phase of any workgroup: {out[9] / 100:.1f} us")

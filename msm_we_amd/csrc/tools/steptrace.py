"""Start times of every training-kernel launch of one bench step, taken in-kernel (no profiler): finds bubbles in the launch stream.
usage: python msm_we_amd/csrc/tools/steptrace.py   (needs _var/lib_steptrace.so: build_var.sh steptrace "-DSTEP_TRACE" train)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from msm_we_amd import _abi, synthetic  # noqa: E402
from msm_we_amd.data import DeviceStore  # noqa: E402

cfg = synthetic.WEConfig("c3", 3, 1000, 10000, 30, 4, 250)
Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, cfg.n_iters)
lib = ctypes.CDLL(os.path.join(ROOT, "msm_we_amd/csrc/_var", "lib_steptrace.so"))
h = _abi.Hamsm(lib=lib)
store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
m = bench.new_model(h, cfg, store, cfg.n_iters)
for _ in range(3):
    bench.one_step(m, cfg, 1337, 0)
h.sync()
N = 32768
lib.hamsm_debug_step_trace(None, None, 1)
bench.one_step(m, cfg, 1337, 0)
h.sync()
out = np.zeros((2, N), dtype=np.uint64)
n = np.zeros(2, dtype=np.uint32)
lib.hamsm_debug_step_trace(out.ctypes.data_as(ctypes.c_void_p), n.ctypes.data_as(ctypes.c_void_p), 0)
print("launches recorded:", n, " timers:", h.timers())
for k, name in ((0, "k_step_assign"), (1, "k_step_update")):
    t = np.sort(out[k, : min(int(n[k]), N)].astype(np.int64)) * 0.01  # us
    d = np.diff(t)
    print(f"{name}: {len(t)} launches over {t[-1] - t[0]:.1f} us; start-to-start median {np.median(d):.2f} us, p90 {np.percentile(d, 90):.2f}, "
          f"p99 {np.percentile(d, 99):.2f}, max {d.max():.2f}")
    big = np.where(d > 25.0)[0]
    print(f"   {len(big)} start-to-start intervals over 25 us, total excess over the median {np.sum(d[big] - np.median(d)):.1f} us; at launches {big[:40].tolist()}")
    print("   intervals:", np.round(d[big[:40]], 1).tolist())
h.close()

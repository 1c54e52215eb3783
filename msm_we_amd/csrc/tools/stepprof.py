"""Timeline of the training-step kernels (k_step_assign / k_step_update), last launch of a C3-shaped run.
usage: python msm_we_amd/csrc/tools/stepprof.py   (needs _var/lib_stepprof{1,2}.so: -DSTEP_PROFILE=1|2)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from msm_we_amd import _abi, synthetic  # noqa: E402
from msm_we_amd.data import DeviceStore  # noqa: E402

cfg = (synthetic.WEConfig("c5t", 5, 24, 50000, 64, 4, 1250) if os.environ.get("HAMSM_PROF_WORKLOAD") == "c5"
       else synthetic.WEConfig("c3s", 3, 60, 10000, 30, 4, 250))
Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, cfg.n_iters)
for level in (1, 2):
    lib = ctypes.CDLL(os.path.join(ROOT, "msm_we_amd/csrc/_var", f"lib_stepprof{level}.so"))
    h = _abi.Hamsm(lib=lib)
    store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
    m = bench.new_model(h, cfg, store, cfg.n_iters)
    for _ in range(2):
        bench.one_step(m, cfg, 1337, 0)
    h.sync()
    prof = np.zeros((2, 4096, 8), dtype=np.uint64)
    lib.hamsm_debug_step_profile(prof.ctypes.data_as(ctypes.c_void_p))
    print(f"== STEP_PROFILE={level}  (100 MHz clock: 1 tick = 10 ns); timers: {h.timers()}")
    for k, name, nslot in ((0, "k_step_assign", 8), (1, "k_step_update", 6))[(1 if os.environ.get("HAMSM_PROF_WORKLOAD") == "c5" else 0):]:
        p = prof[k].astype(np.int64)
        live = p[:, nslot - 1] > p[:, nslot - 1].max() - 2000  # the last launch only (within 20 us of the last end)
        p = p[live]
        t0 = p[:, 0].min()
        start, end = (p[:, 0] - t0) * 0.01, (p[:, nslot - 1] - t0) * 0.01
        print(f"{name}: {live.sum()} workgroups reached the end; starts {start.min():.2f}..{start.max():.2f} us "
              f"(median {np.median(start):.2f}); ends {end.min():.2f}..{end.max():.2f} (median {np.median(end):.2f}); "
              f"in-WG duration median {np.median(end - start):.2f} max {(end - start).max():.2f}")
        np.save(os.path.join(ROOT, "gpurun_out", f"stepprof_l{level}_k{k}.npy"), prof[k])
        if level == 2:
            d = np.diff(p[:, :nslot], axis=1) * 0.01
            print("   phase medians (us):", np.round(np.median(d, axis=0), 2), " p95:", np.round(np.percentile(d, 95, axis=0), 2))
    h.close()

"""phase timers of the persistent training pass (k_train_persist): HAMSM_PERSIST_PROF=1 python persistprof.py [workload] [wpx ...]"""
import ctypes, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", ".."))
os.environ["HAMSM_PERSIST_PROF"] = "1"
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
for wpx in (sys.argv[2:] or ["64"]):
    os.environ["HAMSM_PERSIST_WPX"] = wpx
    import importlib, bench
    importlib.reload(bench)
    from msm_we_amd import _abi
    from msm_we_amd.data import DeviceStore
    cfg = bench.bench_configs()[wl]
    n_it = min(cfg.n_iters, int(os.environ.get("PP_ITERS", "300")))
    h = _abi.Hamsm()
    fn = h.lib.hamsm_debug_persist_profile
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, n_it)
    store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
    m = bench.new_model(h, cfg, store, n_it)
    bench.one_step(m, cfg, 1337, 0)
    out = np.zeros(8, dtype=np.uint64)
    fn(out.ctypes.data_as(ctypes.c_void_p), 1)
    h.timers(reset=True)
    bench.one_step(m, cfg, 1337, 0)
    h.sync()
    fn(out.ctypes.data_as(ctypes.c_void_p), 1)
    t = h.timers()
    steps = max(int(out[6]), 1)
    names = ["tests", "E-step", "wait A", "M-step", "re-seed", "wait B"]
    print(f"wpx={wpx} {wl}: {n_it} iterations, step stage {t['step']*1e3:.2f} ms; per step and stratum (us): " +
          ", ".join(f"{n} {out[i] * 0.01 / steps:.2f}" for i, n in enumerate(names)) + f"; chain {out[7] * 0.01 / cfg.S / 1e3:.2f} ms", flush=True)
    h.close()

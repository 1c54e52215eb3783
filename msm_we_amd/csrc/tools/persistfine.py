"""fine stamps of the persistent pass (library built with -DPERSIST_FINE: csrc/_var/libhamsm_fine.so)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", ".."))
os.environ["HAMSM_PERSIST_PROF"] = "1"
os.environ.setdefault("HAMSM_PERSIST_WPX", "32")
import bench
from msm_we_amd import _abi
from msm_we_amd.data import DeviceStore
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "_var", "libhamsm_fine.so"))
cfg = bench.bench_configs()[sys.argv[1] if len(sys.argv) > 1 else "c3"]
n_it = 300
h = _abi.Hamsm(lib=lib)
fine = lib.hamsm_debug_persist_fine
fine.argtypes = [ctypes.c_void_p]
Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, n_it)
store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
m = bench.new_model(h, cfg, store, n_it)
bench.one_step(m, cfg, 1337, 0)
out = np.zeros(16, dtype=np.uint64)
fine(out.ctypes.data_as(ctypes.c_void_p))
bench.one_step(m, cfg, 1337, 0)
h.sync()
fine(out.ctypes.data_as(ctypes.c_void_p))
n = n_it * cfg.S
names = ["E operands", "E A-load", "E mfma", "E combine", "M pf-issue", "M bitmap", "M members", "M store", "M pf-store"]
print("per step and stratum (us, workgroup rank 0, wave 0; each stamp behind a full s_waitcnt): " +
      ", ".join(f"{nm} {out[i] * 0.01 / n:.2f}" for i, nm in enumerate(names)))

"""Phase stamps of the pipelined pass's M part (library built with -DAHEAD_PROFILE: csrc/_var/lib_apf.so): the last launch's
wave 0 of 64 workgroups, 100 MHz ticks -> microseconds between stamps."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import bench
from msm_we_amd import _abi, synthetic
from msm_we_amd.data import DeviceStore

os.environ.setdefault("HAMSM_TRAIN_AHEAD", "1")
os.environ.setdefault("HAMSM_STEP_LANES", "1")
cfg = synthetic.CONFIGS["c3"]
lib = ctypes.CDLL(os.path.join(ROOT, "msm_we_amd", "csrc", "_var", "lib_apf.so"))
h = _abi.Hamsm(lib=lib)
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 400
Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, n_it)
store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
m = bench.new_model(h, cfg, store, n_it)
for _ in range(2):
    bench.one_step(m, cfg, 1337, 1)
h.sync()
out = np.zeros((64, 8), dtype=np.uint64)
lib.hamsm_debug_ahead_profile.argtypes = [ctypes.c_void_p]
assert lib.hamsm_debug_ahead_profile(out.ctypes.data) == 0
t = out.astype(np.int64)
t0 = t[:, 0].min()
names = ["entry", "round trip 1 + next lists staged", "tests, own rows", "open + undecided rows", "members added", "prefix sums", "member list in LDS", "end"]
order = [0, 1, 2, 3, 5, 6, 4]
cnts = t[:, 7]
print("members of wave 0 per workgroup:", np.sort(cnts & 0xFFFF)[::8], "open rows", int((cnts[0] >> 16) & 0xFFFF), "undecided", int(cnts[0] >> 32))
print("first workgroup entry -> last exit: %.2f us" % ((t[:, 7].max() - t0) / 100.0))
print("entry skew (max - min): %.2f us" % ((t[:, 0].max() - t0) / 100.0))
for a, k in zip(order[:-1], order[1:]):
    d = (t[:, k] - t[:, a]) / 100.0
    ok = (t[:, k] > 0) & (t[:, a] > 0)
    if ok.any():
        print("%-36s median %.2f  max %.2f us" % (names[k], np.median(d[ok]), d[ok].max()))

"""Where the host spends a C3 step: cProfile over N steps (top functions by own time), and the device's idle gaps between
kernels of a step from the timestamps hamsm's stage timers keep.  usage: [HAMSM_PROF_WORKLOAD=c5s] python msm_we_amd/csrc/tools/hostprof.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from msm_we_amd import _abi, synthetic  # noqa: E402
from msm_we_amd.data import DeviceStore  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = synthetic.CONFIGS["c3"]
if os.environ.get("HAMSM_PROF_WORKLOAD") == "c5s":
    cfg = synthetic.WEConfig("c5s", 5, 200, 50000, 64, 4, 1250)   # C5's shape, 10 M frames
h = _abi.Hamsm()
Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, cfg.n_iters)
store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
m = bench.new_model(h, cfg, store, cfg.n_iters)
for _ in range(3):
    bench.one_step(m, cfg, 1337, 1)
h.sync()
h.timers(reset=True)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(steps):
    bench.one_step(m, cfg, 1337, 1)
h.sync()
pr.disable()
dt = time.perf_counter() - t0
print(f"{steps} steps, {dt / steps * 1e3:.3f} ms per step (with cProfile on)")
print("timers:", h.timers())
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
t0 = time.perf_counter()
for _ in range(steps):
    bench.one_step(m, cfg, 1337, 1)
h.sync()
print(f"without cProfile: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")

# ---- the library calls of one step on the host's clock: name, enter, exit; what lies between them is Python ----
log = []
lib = h.lib
for name in [n for n in list(vars(lib)) if n.startswith("hamsm_")]:
    fn = getattr(lib, name)

    def wrap(*a, _fn=fn, _name=name):
        t0 = time.perf_counter()
        r = _fn(*a)
        log.append((_name, t0, time.perf_counter()))
        return r
    setattr(lib, name, wrap)
for _ in range(2):
    bench.one_step(m, cfg, 1337, 1)
h.sync()
log.clear()
t_begin = time.perf_counter()
bench.one_step(m, cfg, 1337, 1)
h.sync()
t_end = time.perf_counter()
print(f"\none step with the call log on: {(t_end - t_begin) * 1e3:.3f} ms; {len(log)} library calls")
prev = t_begin
inside = 0.0
for name, a, b in log:
    inside += b - a
    if (a - prev) > 20e-6 or (b - a) > 50e-6:
        print(f"  +{(a - t_begin) * 1e3:8.3f} ms  python before {1e6 * (a - prev):7.0f} us   {name:34s} {1e6 * (b - a):8.0f} us")
    prev = b
print(f"  python after the last call {1e6 * (t_end - prev):.0f} us; inside the library {inside * 1e3:.3f} ms, outside {(t_end - t_begin - inside) * 1e3:.3f} ms")

"""Wall-clock trace of one bench step: every call into the library with its host-side duration, in order."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import bench
from msm_we_amd import _abi, synthetic
from msm_we_amd.data import DeviceStore

_wl = os.environ.get("HAMSM_TRACE_WORKLOAD", "c3")
cfg = synthetic.WEConfig("c5s", 5, 200, 50000, 64, 4, 1250) if _wl == "c5s" else synthetic.CONFIGS[_wl]
h = _abi.Hamsm()
Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, cfg.n_iters)
store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
m = bench.new_model(h, cfg, store, cfg.n_iters)
for _ in range(2):
    bench.one_step(m, cfg, 1337, 0)
h.sync()
log = []
names = [n for n in dir(h) if not n.startswith("_") and callable(getattr(h, n)) and n not in ("sync",)]
for n in names:
    f = getattr(h, n)
    def wrap(f=f, n=n):
        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            log.append((n, t0, time.perf_counter()))
            return r
        return g
    try:
        setattr(h, n, wrap())
    except Exception:
        pass
T0 = time.perf_counter()
bench.one_step(m, cfg, 1337, 0)
h.sync()
T1 = time.perf_counter()
print(f"step wall {1e3 * (T1 - T0):.2f} ms; calls:")
prev = T0
agg = {}
for n, a, b in log:
    agg.setdefault(n, [0, 0.0]); agg[n][0] += 1; agg[n][1] += b - a
    if b - a > 5e-5 or a - prev > 5e-5:
        print(f"  +{1e3 * (a - T0):7.2f} ms  gap {1e3 * (a - prev):6.2f}  {n:28s} {1e3 * (b - a):7.3f} ms")
    prev = b
print("totals:", {k: (v[0], round(1e3 * v[1], 2)) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]})
print(f"time inside library calls {1e3 * sum(v[1] for v in agg.values()):.2f} ms")

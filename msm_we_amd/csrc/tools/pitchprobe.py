"""What a 128-byte row pitch would buy at C3's shape: the bench step with 30 features per frame (120-byte rows, the product) and
with 32 (128-byte rows: every row one cache line, every operand load aligned).  Stage timers per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench
from msm_we_amd import _abi, synthetic
from msm_we_amd.data import DeviceStore

for D in (30, 32, 30, 32):
    cfg = synthetic.WEConfig("c3", 3, 1000, 10000, D, 4, 250)
    Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, cfg.n_iters)
    h = _abi.Hamsm()
    store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
    m = bench.new_model(h, cfg, store, cfg.n_iters)
    for _ in range(2):
        bench.one_step(m, cfg, 1337, 0)
    h.timers(reset=True)
    h.sync()
    t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        bench.one_step(m, cfg, 1337, 0)
    h.sync()
    dt = (time.perf_counter() - t0) / reps
    tm = {k: round(v / reps * 1e3, 3) for k, v in h.timers().items()}
    print(f"D={D}: {dt * 1e3:.2f} ms per step; {tm}", flush=True)
    h.close()
    del store, m

"""Where the ranked E-step starts to pay: training-step time of a 100-iteration pass with the f64 E-step (HAMSM_STEP_RANKED=0) and
with the ranked one forced (=2) over shapes between C3's (0.07 GFLOP per launch of two strata) and C5's (4 GFLOP)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench
from msm_we_amd import _abi, synthetic
from msm_we_amd.data import DeviceStore

for D, K, segs in ((30, 250, 10000), (64, 250, 20000), (64, 500, 20000), (32, 1000, 20000), (64, 1000, 20000), (64, 1250, 50000)):
    cfg = synthetic.WEConfig("rg", 5, 100, segs, D, 4, K)
    Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, cfg.n_iters)
    res = {}
    for mode in (0, 2):
        h = _abi.Hamsm()
        fn = h.lib.hamsm_debug_rank_stats
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        fn(h.ctx, None, 1, mode)
        store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
        m = bench.new_model(h, cfg, store, cfg.n_iters)
        for _ in range(2):
            bench.one_step(m, cfg, 1337, 0)
        h.timers(reset=True)
        for _ in range(3):
            bench.one_step(m, cfg, 1337, 0)
        h.sync()
        res[mode] = h.timers()["step"] / 3 * 1e3
        h.close()
    gf = 2 * 2.0 * (segs / 4) * K * D / 1e9
    print(f"D={D} K_s={K} rows/call={segs // 4}: {gf:.2f} GFLOP per launch of two strata; training steps f64 {res[0]:.2f} ms, ranked {res[2]:.2f} ms per 100-iteration pass", flush=True)

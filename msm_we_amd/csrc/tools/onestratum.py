"""One stratum's training chain alone on the chip (what a rank does under train_shards = "strata" with N >= S): the training-step
time of a pass over 1000 iterations with ONE stratum of C3's / C5's per-stratum shape, next to the four-strata pass of the bench."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench
from msm_we_amd import _abi, synthetic
from msm_we_amd.data import DeviceStore

for name, cfg in (("C3, one stratum (2 500 segments x 30, 250 clusters)", synthetic.WEConfig("o3", 3, 1000, 2500, 30, 1, 250)),
                  ("C3, four strata", synthetic.WEConfig("c3", 3, 1000, 10000, 30, 4, 250)),
                  ("C5 shape, one stratum (12 500 x 64, 1250 clusters), 200 iterations", synthetic.WEConfig("o5", 5, 200, 12500, 64, 1, 1250)),
                  ("C5 shape, four strata, 200 iterations", synthetic.WEConfig("c5s", 5, 200, 50000, 64, 4, 1250))):
    Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, cfg.n_iters)
    h = _abi.Hamsm()
    store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
    m = bench.new_model(h, cfg, store, cfg.n_iters)
    for _ in range(2):
        bench.one_step(m, cfg, 1337, 0)
    h.timers(reset=True)
    reps = 5
    for _ in range(reps):
        bench.one_step(m, cfg, 1337, 0)
    h.sync()
    t = h.timers()
    print(f"{name}: training steps {t['step'] / reps * 1e3:.2f} ms per pass = {t['step'] / reps / (cfg.n_iters - 1) * 1e6:.1f} us per call; seeding {t['seed'] / reps * 1e3:.2f} ms", flush=True)
    h.close()

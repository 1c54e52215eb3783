import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from msm_we_amd import _abi, synthetic
from msm_we_amd.data import DeviceStore
shapes = {"tiny": synthetic.WEConfig("tiny", 9, 300, 2000, 30, 4, 50), "s8x64": synthetic.WEConfig("s8", 9, 300, 8000, 30, 8, 64),
          "s16x32_d10": synthetic.WEConfig("s16", 9, 300, 8000, 10, 16, 32), "c3": synthetic.CONFIGS["c3"]}
for name, cfg in shapes.items():
    n_it = 300
    Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, n_it)
    res = {}
    for persist in ("0", "1"):
        os.environ["HAMSM_TRAIN_PERSIST"] = persist
        h = _abi.Hamsm()
        store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
        m = bench.new_model(h, cfg, store, n_it)
        bench.one_step(m, cfg, 1337, 0)
        h.timers(reset=True)
        for _ in range(3):
            bench.one_step(m, cfg, 1337, 0)
        h.sync()
        res[persist] = h.timers()["step"] / 3 * 1e3
        lab = np.concatenate(m.dtrajs)
        res["lab" + persist] = lab
        h.close()
    print(f"{name}: S={cfg.S} K_s={cfg.K_s} D={cfg.D} segs={cfg.segs_per_iter}: multi-kernel {res['0']:.2f} ms, persistent {res['1']:.2f} ms per {n_it}-iteration pass; labels equal: {np.array_equal(res['lab0'], res['lab1'])}", flush=True)

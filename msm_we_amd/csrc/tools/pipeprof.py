"""cProfile of the post-clustering part of build_analyze_model (organize, T, solve, block validation) on C3 (or the workload named in argv[1], e.g. c5s)."""
import cProfile, os, pstats, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import bench
from msm_we_amd import _abi, synthetic
from msm_we_amd.data import DeviceStore

cfg = bench.bench_configs()[sys.argv[1] if len(sys.argv) > 1 else "c3"]
h = _abi.Hamsm()
Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, cfg.n_iters)
store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
m = bench.new_model(h, cfg, store, cfg.n_iters)
bench.full_pipeline(m, cfg, 1337, 0)
pr = cProfile.Profile()
pr.enable()
r = bench.full_pipeline(m, cfg, 1337, 0)
pr.disable()
print(r["seconds"])
pstats.Stats(pr).sort_stats("tottime").print_stats(40)

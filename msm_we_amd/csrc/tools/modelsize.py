"""which attributes of the post-clustering model make its deep copies (block validation) expensive?"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from msm_we_amd import _abi, synthetic
from msm_we_amd.data import DeviceStore

cfg = bench.bench_configs()[sys.argv[1] if len(sys.argv) > 1 else "c3"]
h = _abi.Hamsm()
Xp, Xc, p0, p1, w, offs = bench.make_shard(cfg, 1, cfg.n_iters)
store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
m = bench.new_model(h, cfg, store, cfg.n_iters)
bench.full_pipeline(m, cfg, 1337, 0)


def size(v, seen):
    if id(v) in seen:
        return 0
    seen.add(id(v))
    if isinstance(v, np.ndarray):
        return v.nbytes
    if isinstance(v, (list, tuple)):
        return sum(size(x, seen) for x in v)
    if isinstance(v, dict):
        return sum(size(x, seen) for x in v.values())
    if hasattr(v, "__dict__") and not isinstance(v, type):
        return sum(size(x, seen) for k, x in v.__dict__.items() if k not in ("_h", "_store", "_source", "model"))
    return 0


pc = m.post_cluster_model
rows = sorted(((size(v, set()), k) for k, v in pc.__dict__.items()), reverse=True)
for b, k in rows[:12]:
    print(f"{b / 1e6:10.2f} MB  {k}")

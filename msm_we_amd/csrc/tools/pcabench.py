"""hamsm_pca_accumulate throughput: rows x features resident in HBM, one accumulate call each; prints ms, rows/s, the f64
flops the chip executed (2 x Dp^2/2-ish per row) and the HBM rate of the feature stream.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from msm_we_amd import _abi

h = _abi.Hamsm()
for D, n, dt in ((30, 10_000_000, np.float32), (64, 5_000_000, np.float32), (80, 2_000_000, np.float64), (816, 240_000, np.float32)):
    rng = np.random.default_rng(0)
    X = rng.normal(size=(min(n, 500_000), D)).astype(dt)
    X = np.tile(X, (n // len(X), 1))
    n = len(X)
    d = h.to_device(X)
    code = _abi.F32 if dt == np.float32 else _abi.F64
    h.pca_begin(D)
    h.pca_accumulate(d, code, D, n)
    h.sync()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        h.pca_accumulate(d, code, D, n)
    h.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    h.pca_finish()
    Dp = (D + 1 + 31) // 32 * 32
    nb = Dp // 32
    tiles = nb * 3 + (nb * (nb - 1) // 2) * 4            # 16x16 tiles computed per 4 rows
    fl = tiles * 2048 * (n / 4)
    print(f"D={D} {np.dtype(dt).name} n={n}: {ms:.3f} ms = {n / ms / 1e3:.1f} M rows/s; executed {fl / ms / 1e9:.1f} TFLOP/s f64 "
          f"(algorithmic D(D+1) flops/row: {n * D * (D + 1) / ms / 1e9:.1f}); feature stream {X.nbytes / ms / 1e6:.0f} GB/s", flush=True)
    d.free()
h.close()

# model level: dimReduce("pca") on a C3-shaped slice (100 iterations x 10 000 segments x 30 features), host IncrementalPCA
# (the reference's estimator, default) against the device fit
from msm_we_amd import synthetic
from msm_we_amd.data import InMemorySource
from msm_we_amd.model import modelWE

cfg = synthetic.WEConfig("p", 3, 100, 10000, 30, 4, 250)
lat = synthetic.latent_centres(cfg)
iters = {i: synthetic.make_iteration(cfg, i, lat) for i in range(1, cfg.n_iters + 1)}
h = _abi.Hamsm()
res = {}
for mode in ("host", "device", "device"):
    m = modelWE(h=h)
    m.pca_fit = mode
    m.set_iteration_source(InMemorySource(iters))
    basis, target = synthetic.basis_target_bounds(cfg.S)
    m.initialize(["x"], {"coords": None, "nAtoms": cfg.D, "coord_ndim": 1}, "t", basis_pcoord_bounds=basis, target_pcoord_bounds=target,
                 tau=1.0, dim_reduce_method="pca")
    m.get_iterations()
    t0 = time.perf_counter()
    m.dimReduce()
    res[mode] = (time.perf_counter() - t0, m.ndim, m.coordinates.components_)
s = np.linalg.svd(res["host"][2] @ res["device"][2].T, compute_uv=False)
print(f"dimReduce('pca'), 100 x 10000 x 30: host IncrementalPCA {res['host'][0]:.3f} s ({res['host'][1]} components), device fit "
      f"{res['device'][0]:.3f} s ({res['device'][1]} components); smallest principal cosine {s.min():.6f}", flush=True)
h.close()

#!/usr/bin/env python3
"""bench.py -- haMSM construction throughput on MI355X: frames/s clustered + flux-matrix built.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (default ``c3`` = BASELINE.json configs[2], the configuration the metric is quoted on):
synthetic WE data, 1000 iterations x 10 000 segments, 30 float32 features per frame, 4 strata x 250
clusters (K = 1000).  One *step* = one pass of the hot path over the resident shard:

    stratified mini-batch k-means training (999 iterations, sklearn-exact control flow and draws)
    -> assignment of the parent and child frame of every segment (2 x 10 M rows)
    -> flux-matrix accumulation over the segments (the (K+2)^2 matrix stays in HBM)

Inputs are generated on the host, uploaded once, and are resident in HBM when the timed region
starts.  N > 1, default = BASELINE config 4 (strong scaling): the workload's 10 M frames are split over
the ranks by WE iteration; the training frames of every rank reach rank 0 once at staging, which trains
on all of them and broadcasts the model with RCCL (where a stratum's training call is matrix-bound --
config 5's shape -- they reach every rank instead and stratum s trains on rank s % N: the strata's
chains of partial_fits are independent); every rank discretizes its shard and the partial flux matrices
are all-reduced over xGMI -- N ranks produce the model of one process on the same data.  ``--scaling weak``: every rank its
own full shard, rank 0 trains on its block (the reference's ``iters_to_use``).

Rank 0 prints ONE JSON line (contract in the task statement) with ``roofline`` for the dominant kernel
of the step -- the E-step of the training pass: ``k_step_assign`` (f64 matrix unit) at C3,
``k_step_assign_ranked`` (bf16 matrix unit) at C5's shape -- and ``cpu_baseline`` (the oracle = scalar C
port, bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from msm_we_amd import _abi, synthetic  # noqa: E402
from msm_we_amd.data import DeviceStore  # noqa: E402
from msm_we_amd.model import modelWE  # noqa: E402

PEAK_F64_MFMA_TFLOPS = 78.6   # MI355X FP64 matrix (= vector) peak, dense (vendor figure)
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X FP32 matrix peak, dense (MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 155 TF measured)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X BF16 matrix peak, dense (2.2 PFLOP/s measured here: csrc/tools/mfma_f32_rate.hip)
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec


class Mapper:
    """attribute-only RectilinearBinMapper (the library does the assignment)"""

    def __init__(self, boundaries):
        self.boundaries = [np.asarray(b, dtype=np.float32) for b in boundaries]
        self.nbins = int(np.prod([len(b) - 1 for b in self.boundaries]))


Mapper.__name__ = "RectilinearBinMapper"


def bench_configs():
    cfgs = dict(synthetic.CONFIGS)
    cfgs["c5s"] = synthetic.WEConfig("c5s", 5, 200, 50000, 64, 4, 1250)   # C5 shape, 10 M frames (single-GPU slice)
    cfgs["tiny"] = synthetic.WEConfig("tiny", 9, 50, 2000, 30, 4, 50)
    return cfgs


def make_shard(cfg, it_first, n_iters, log=None):
    """iterations it_first .. it_first+n_iters-1 as host arrays (float32 features)"""
    lat = synthetic.latent_centres(cfg)
    n = n_iters * cfg.segs_per_iter
    Xp = np.empty((n, cfg.D), np.float32)
    Xc = np.empty((n, cfg.D), np.float32)
    p0 = np.empty((n, 1))
    p1 = np.empty((n, 1))
    w = np.empty(n)
    t0 = time.time()
    for k in range(n_iters):
        d = synthetic.make_iteration(cfg, it_first + k, lat)
        a, b = k * cfg.segs_per_iter, (k + 1) * cfg.segs_per_iter
        Xp[a:b], Xc[a:b], p0[a:b], p1[a:b], w[a:b] = d["Xp"], d["Xc"], d["p0"], d["p1"], d["w"]
        if log and k and k % 250 == 0:
            log(f"generated {k}/{n_iters} iterations ({time.time() - t0:.1f}s)")
    offs = np.arange(n_iters + 1, dtype=np.int64) * cfg.segs_per_iter
    return Xp, Xc, p0, p1, w, offs


def new_model(h, cfg, store, n_iters):
    m = modelWE(h=h)
    m.pcoord_ndim = 1
    basis, target = synthetic.basis_target_bounds(cfg.S)
    m.basis_pcoord_bounds = basis
    m.target_pcoord_bounds = target
    m.tau = 1.0
    # n_iters iterations are resident and complete: the reference's maxIter counts iterations that have a
    # successor (_data.py:860-866), and trains / discretizes range(1, maxIter)
    m.attach_store(store, maxIter=None)
    m.maxIter = n_iters + 1
    m.numSegments = np.diff(store.offsets).astype(float)
    return m


def one_step(m, cfg, seed, flux_mode):
    bm = Mapper([synthetic.strata_boundaries(cfg.S)])
    m.flux_mode = flux_mode
    m.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=seed)
    m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter)
    return m


ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libhamsm_oracle.so")
_W = {}  # per-process state of the CPU pool workers


def _cpu_pool_init(cfg_tuple, it_first, n_it, cents, remap):
    """pool worker, untimed: an oracle context with the trained model and this worker's block of iterations"""
    cfg = synthetic.WEConfig(*cfg_tuple)
    h = _abi.Hamsm(lib=ctypes.CDLL(ORACLE_SO))
    h.set_geometry(cfg.D, 1, cfg.S)
    h.set_mapper_rectilinear([synthetic.strata_boundaries(cfg.S)])
    basis, target = synthetic.basis_target_bounds(cfg.S)
    h.set_bounds(basis, target)
    for s, (Cm, csq) in enumerate(cents):
        h.set_centroids(s, Cm, csq=csq)
    h.set_remap(np.asarray(remap, dtype=np.int32))
    Xp, Xc, p0, p1, w, offs = make_shard(cfg, it_first, n_it)
    _W.update(h=h, cfg=cfg, d=(h.to_device(Xp), h.to_device(Xc), h.to_device(p0), h.to_device(p1), h.to_device(w)), n=len(w))
    return n_it


def _cpu_pool_run(_):
    """pool worker, timed: discretize the parent and child frames of the block and accumulate its flux matrix"""
    h, cfg, (dXp, dXc, dp0, dp1, dw), n = _W["h"], _W["cfg"], _W["d"], _W["n"]
    K = cfg.K_s * cfg.S
    t0 = time.perf_counter()
    la, lb = h.alloc(n, np.int32), h.alloc(n, np.int32)
    h.assign(dXp, 0, cfg.D, dp0, n, la)
    h.assign(dXc, 0, cfg.D, dp1, n, lb)
    dF = h.zeros((K + 2) * (K + 2), np.float64)
    h.flux_accumulate(la, lb, dp0, dp1, dw, n, K, dF, 0)
    F = dF.to_host()
    return time.perf_counter() - t0, float(F.sum())


def reference_shaped_leg(cfg, seed, n_train=4):
    """The reference's CALL PATTERN with its own estimator (scikit-learn), on one core: one MiniBatchKMeans.partial_fit
    per WE bin per iteration (_clustering.py:843-899), then for ONE iteration the per-frame loop of
    StratifiedClusters.predict -- offsets recomputed and a single-row `predict([coord])` for every parent and child frame
    (stratified_clustering.py:152-196) -- and the per-iteration flux accumulation (_fluxmatrix.py:97-164).  Returns frames/s
    per core or None without scikit-learn."""
    try:
        from sklearn.cluster import MiniBatchKMeans
    except Exception:  # noqa: BLE001
        return None
    lat = synthetic.latent_centres(cfg)
    its = [synthetic.make_iteration(cfg, it, lat) for it in range(1, n_train + 1)]
    S, K_s = cfg.S, cfg.K_s
    basis, target = synthetic.basis_target_bounds(S)
    edges = np.asarray(synthetic.strata_boundaries(S), dtype=np.float32)

    def bins(p):
        return np.digitize(p[:, 0].astype(np.float32), edges) - 1

    def in_box(p, box):
        return (p[:, 0] > box[0][0]) & (p[:, 0] < box[0][1])

    models = [MiniBatchKMeans(n_clusters=K_s, random_state=seed, n_init=1) for _ in range(S)]
    t0 = time.perf_counter()
    for d in its:
        ok = ~(in_box(d["p0"], basis) | in_box(d["p0"], target))
        b = bins(d["p0"])
        for s in range(S):
            rows = np.nonzero(ok & (b == s))[0]
            if len(rows) >= K_s:
                models[s].partial_fit(d["Xc"][rows].astype(np.float64))
    t_train = (time.perf_counter() - t0) / n_train
    d = its[-1]
    K = S * K_s
    t0 = time.perf_counter()
    lab = []
    for X, p in ((d["Xp"], d["p0"]), (d["Xc"], d["p1"])):
        b = bins(p)
        tgt, bas = in_box(p, target), in_box(p, basis)
        out = np.empty(len(X), dtype=np.int64)
        Xd = X.astype(np.float64)
        for i in range(len(X)):
            if tgt[i]:
                out[i] = K + 1
            elif bas[i]:
                out[i] = K
            else:
                off = sum(len(models[q].cluster_centers_) for q in range(b[i]))
                out[i] = models[b[i]].predict([Xd[i]])[0] + off
        lab.append(out)
    F = np.zeros((K + 2, K + 2))
    np.add.at(F, (lab[0], lab[1]), d["w"])
    t_iter = time.perf_counter() - t0
    n = len(d["w"])
    return {"value": n / (t_train + t_iter), "cores": 1,
            "how": f"scikit-learn, the reference's call pattern: {n_train} iterations of per-bin partial_fit ({t_train:.2f} s per "
                   f"iteration) + one iteration of per-frame single-row predict for {2 * n} frames and its flux ({t_iter:.1f} s)"}


def cpu_baseline_worker(workload, n_iters_sample, seed, cores):
    """Runs in a child process that never touches the GPU (it forks a pool).  (i) the oracle = scalar C restatement on
    one core over the first `n_iters_sample` iterations; (ii) the same with the reference's parallel structure:
    training sequential in the driver (it is, _clustering.py:664-716), discretization and flux matrix over `cores`
    worker processes by iteration block (the Ray tasks of _clustering.py:1183 / _fluxmatrix.py:275)."""
    import multiprocessing as mp

    cfg = bench_configs()[workload]
    h = _abi.Hamsm(lib=ctypes.CDLL(ORACLE_SO))
    Xp, Xc, p0, p1, w, offs = make_shard(cfg, 1, n_iters_sample)
    store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
    m = new_model(h, cfg, store, n_iters_sample)
    t_disc = [0.0]
    inner = m.launch_ray_discretization

    def timed_disc(*a, **k):
        t = time.perf_counter()
        r = inner(*a, **k)
        t_disc[0] += time.perf_counter() - t
        return r

    m.launch_ray_discretization = timed_disc
    t0 = time.perf_counter()
    one_step(m, cfg, seed, 1)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter)
    t_flux = time.perf_counter() - t0
    t_train = dt - t_disc[0] - t_flux
    frames = n_iters_sample * cfg.segs_per_iter
    out = {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"first {n_iters_sample} of {cfg.n_iters} iterations ({frames} segments), same train+assign+flux "
                     f"step, oracle/hamsm_oracle.c single thread, {dt:.1f} s"}
    if cores > 1:
        cents = [h.get_centroids(s) for s in range(cfg.S)]
        cents = [(c[0], c[2]) for c in cents]
        remap = [m.clusters.we_remap[b] for b in range(cfg.S)]
        per = -(-n_iters_sample // cores)
        blocks = [(1 + k * per, min(per, n_iters_sample - k * per)) for k in range(cores) if k * per < n_iters_sample]
        cfg_t = (cfg.name, cfg.seed, cfg.n_iters, cfg.segs_per_iter, cfg.D, cfg.S, cfg.K_s)
        with mp.get_context("fork").Pool(len(blocks)) as pool:
            # one task per worker: chunksize 1 and as many tasks as processes
            res = [pool.apply_async(_cpu_pool_init, (cfg_t, a, n, cents, remap)) for a, n in blocks]
            [r.get() for r in res]
            t0 = time.perf_counter()
            res = pool.map(_cpu_pool_run, range(len(blocks)), chunksize=1)
            t_par = time.perf_counter() - t0
        out["all_cores"] = {"value": frames / (t_train + t_par), "cores": len(blocks),
                            "how": f"training on one core ({t_train:.2f} s, sequential in the reference too) + discretization "
                                   f"and flux over {len(blocks)} processes by iteration block ({t_par:.2f} s wall)"}
    if os.environ.get("HAMSM_BENCH_REFERENCE_SHAPED", "1") != "0" and cfg.segs_per_iter * 2 <= 40000:
        rs = reference_shaped_leg(cfg, seed)
        if rs is not None:
            out["reference_shaped"] = rs
    print(json.dumps(out), flush=True)


def cpu_baseline(workload, n_iters_sample, seed, cores):
    """child process: the CPU legs fork a process pool, which must not happen in a process that holds a GPU context"""
    import subprocess

    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", workload, str(n_iters_sample), str(seed),
                        str(cores)], capture_output=True, text=True, check=True,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1"))
    return json.loads(r.stdout.strip().splitlines()[-1])


def torch_exchange(h, dist, torch, rank, world, dev="cuda"):
    """Stand-in for the library's RCCL entry points (only used when its communicator cannot be created): the same
    collectives on the same device buffers through torch.distributed's NCCL(=RCCL) process group, staged via the host."""
    h.nranks, h.rank = world, rank

    def allreduce(dtype, op):
        def f(d_buf, count):
            a = np.empty(count, dtype=dtype)
            h.d2h(a, d_buf)
            t = torch.from_numpy(a).to(dev)
            dist.all_reduce(t, op=op)
            h.h2d(d_buf, t.cpu().numpy())
        return f

    h.allreduce_sum_f64 = allreduce(np.float64, dist.ReduceOp.SUM)
    h.allreduce_sum_i64 = allreduce(np.int64, dist.ReduceOp.SUM)
    h.allreduce_min_f64 = allreduce(np.float64, dist.ReduceOp.MIN)
    h.allreduce_max_f64 = allreduce(np.float64, dist.ReduceOp.MAX)

    def bcast(d_buf, nbytes, root=0):
        a = np.empty(nbytes, dtype=np.uint8)
        h.d2h(a, d_buf)
        t = torch.from_numpy(a).to(dev)
        dist.broadcast(t, root)
        h.h2d(d_buf, t.cpu().numpy())

    def bcast_model(root=0):
        state = [[h.get_centroids(s) for s in range(h.S)] if rank == root else None]
        dist.broadcast_object_list(state, src=root)
        if rank != root:
            for s, (Cm, counts, csq) in enumerate(state[0]):
                h.set_centroids(s, Cm if len(Cm) else None, csq=csq if len(Cm) else None, counts=counts if len(Cm) else None)

    def gather_to_root(d_send, send_bytes, d_recv, recv_bytes, root=0):
        a = np.empty(int(send_bytes), dtype=np.uint8)
        h.d2h(a, d_send)
        parts = [None] * world if rank == root else None
        dist.gather_object(a, parts, dst=root)
        if rank == root:
            h.h2d(d_recv, np.concatenate(parts))

    def bcast_strata(roots):
        for s in range(h.S):
            root = int(roots[s])
            state = [h.get_centroids(s) if rank == root else None]
            dist.broadcast_object_list(state, src=root)
            if rank != root:
                Cm, counts, csq = state[0]
                h.set_centroids(s, Cm if len(Cm) else None, csq=csq if len(Cm) else None, counts=counts if len(Cm) else None)

    def exchange(d_send, send_bytes, d_recv, recv_bytes):
        sb, rb = [int(x) for x in send_bytes], [int(x) for x in recv_bytes]
        a = np.empty(sum(sb), dtype=np.uint8)
        if a.size:
            h.d2h(a, d_send)
        cuts = np.cumsum([0] + sb)
        rows = [None] * world
        dist.all_gather_object(rows, [a[cuts[q]: cuts[q + 1]] for q in range(world)])
        mine = [rows[r][rank] for r in range(world)]
        assert [len(x) for x in mine] == rb
        if sum(rb):
            h.h2d(d_recv, np.concatenate(mine))

    h.exchange = exchange
    h.bcast, h.bcast_model, h.gather_to_root, h.bcast_strata = bcast, bcast_model, gather_to_root, bcast_strata
    h.comm_destroy = lambda: None


def full_pipeline(m, cfg, seed, flux_mode):
    """The reference's ``build_analyze_model`` sequence from clustering onwards (msm_we.py:588-882): train + discretize,
    flux matrix, cleaning (re-discretize + re-flux), T as CSR on the device, steady state + target flux on the host,
    then 2-group / 4-block validation (per group: flux, clean, T, solve).  Timed once, outside the timed steps."""
    bm = Mapper([synthetic.strata_boundaries(cfg.S)])
    m.flux_mode = flux_mode
    t = {}

    def lap(name, fn):
        t0 = time.perf_counter()
        fn()
        m.h.sync()
        t[name] = time.perf_counter() - t0

    lap("cluster", lambda: m.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm,
                                                  random_state=seed, store_validation_model=True))
    lap("fluxmatrix", lambda: m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter, use_ray=True))
    lap("organize", lambda: m.organize_fluxMatrix(use_ray=True))
    lap("tmatrix_csr", m.get_Tmatrix)
    lap("steady_state", m.get_steady_state)
    lap("target_flux", m.get_steady_state_target_flux)
    lap("block_validation", lambda: m.do_block_validation(cross_validation_groups=2, cross_validation_blocks=4, use_ray=True))
    total = sum(t.values())
    return {"seconds": {k: round(v, 5) for k, v in t.items()}, "total_seconds": total, "states": int(m.Tmatrix.shape[0]),
            "tmatrix_nnz": int(m.Tmatrix_csr.nnz), "JtargetSS": float(m.JtargetSS),
            "validation_JtargetSS": [float(v.JtargetSS) for v in m.validation_models]}


def _c2_setup(h, cfg, d_train):
    """config 2's model: one stratum, 200 centroids "from a 20 k-row partial_fit"  -- the product's own
    k-means++ + one mini-batch step on the first 20 iterations' child frames"""
    from msm_we_amd.stratified_clustering import StratifiedClusters

    h.set_geometry(cfg.D, 1, cfg.S)
    h.set_mapper_rectilinear([synthetic.strata_boundaries(cfg.S)])
    basis, target = synthetic.basis_target_bounds(cfg.S)
    h.set_bounds(basis, target)
    cl = StratifiedClusters(Mapper([synthetic.strata_boundaries(cfg.S)]), None, cfg.K_s, h=h, random_state=2)
    cl.cluster_models[0].partial_fit(d_train)
    cl._activate()
    return cl


def c2_cpu_worker(n_rows):
    """child process, no GPU: the oracle's `hamsm_assign` (scalar C, one core) over config 2's first `n_rows` child frames"""
    cfg = bench_configs()["c2"]
    h = _abi.Hamsm(lib=ctypes.CDLL(ORACLE_SO))
    _, Xc, _, p1, _, _ = make_shard(cfg, 1, -(-n_rows // cfg.segs_per_iter))
    Xc, p1 = Xc[:n_rows], p1[:n_rows]
    _c2_setup(h, cfg, Xc[:20000])
    d_X, d_p, d_l = h.to_device(Xc), h.to_device(p1), h.alloc(n_rows, np.int32)
    t0 = time.perf_counter()
    h.assign(d_X, 0, cfg.D, d_p, n_rows, d_l)
    dt = time.perf_counter() - t0
    print(json.dumps({"value": n_rows / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                      "sample": f"the first {n_rows} of config 2's 1 000 000 child frames, same assign pass, oracle/hamsm_oracle.c "
                                f"single thread, {dt:.1f} s", "label_checksum": int(d_l.to_host().astype(np.int64).sum())}), flush=True)


def measure_c2(device, args, log, reps=1, cpu=True):
    """BASELINE configs[1]: synthetic 1 M frames x 10 features, 200 clusters, 1 stratum -- distance / assign only.  A step = one
    `hamsm_assign` over the resident frames (bucket pass + ranking pass + exact pass on what the ranking cannot certify; labels
    stay in HBM), i.e. what `StratifiedClusters.predict` (stratified_clustering.py:152-196) computes for these frames.
    `reps` > 1: the same 1 M frames resident `reps` times over (64 M frames: the per-launch fixed costs amortised -- the kernel's
    asymptote).  Algorithmic bytes per frame: D*4 + 8*ndim + 4 = 52; flops 2*K*D = 4000."""
    cfg = bench_configs()["c2"]
    steps, warmup = max(args.steps, 20), max(args.warmup, 5)
    h = _abi.Hamsm(lib=ctypes.CDLL(args.lib) if args.lib else None, device=device)
    try:
        _, Xc, _, p1, _, _ = make_shard(cfg, 1, cfg.n_iters if not args.c2_frames else -(-args.c2_frames // cfg.segs_per_iter))
        if args.c2_frames:
            Xc, p1 = Xc[: args.c2_frames], p1[: args.c2_frames]
        n1 = len(p1)
        _c2_setup(h, cfg, Xc[:20000])
        n = n1 * reps
        d_X, d_p, d_l = h.alloc((n, cfg.D), np.float32), h.alloc((n, 1), np.float64), h.alloc(n, np.int32)
        for r in range(reps):
            h.h2d(d_X.offset(r * n1 * cfg.D), Xc)
            h.h2d(d_p.offset(r * n1), p1)
        for _ in range(warmup):
            h.assign(d_X, 0, cfg.D, d_p, n, d_l)
        h.sync()
        h.timers(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            h.assign(d_X, 0, cfg.D, d_p, n, d_l)
        h.sync()
        elapsed = time.perf_counter() - t0
        timers = h.timers()
        lab = d_l.to_host()
        # the labels of every repeat are those of the first (the kernel buckets frames per chunk: position must not matter)
        if reps > 1 and not all(np.array_equal(lab[:n1], lab[r * n1: (r + 1) * n1]) for r in range(1, reps)):
            raise SystemExit("[bench] c2: labels differ between repeats of the same frames")
        checksum = int(lab[:n1].astype(np.int64).sum())
        # the reference's post-condition: labels on the host
        t0 = time.perf_counter()
        for _ in range(3):
            h.assign(d_X, 0, cfg.D, d_p, n, d_l)
            a = _abi.HostBlock.array(h.lib, n, np.int32)
            h.d2h(a, d_l)
        h.sync()
        step_labels_s = (time.perf_counter() - t0) / 3
        del a
        k_ms = timers["assign"] * 1e3 / steps           # ranking + exact kernels per step (HIP events on the library's stream)
        o_ms = timers["other"] * 1e3 / steps            # the bucket passes of the same call
        bytes_alg = (cfg.D * 4 + 8 + 4) * n
        flops = 2.0 * cfg.K_s * cfg.D * n
        gbs = bytes_alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else float("nan")
        tf = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else float("nan")
        # which roof binds: 4000 flop / 52 B = 77 flop/B; the ranking pass runs on the bf16 matrix unit with 3 piece products per
        # f32 product (ridge 2500/3 TF / 8 TB/s = 104 flop/B) -> below the ridge: HBM is the roof of this shape
        out = {"metric": "frames/sec assigned (distance / assign only)", "value": n * steps / elapsed, "unit": "frames/s",
               "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": elapsed * 1e3 / steps, "higher_is_better": True,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"c2: {n} frames ({reps} x 1000 WE iterations x 1000 segments) x {cfg.D} f32 features, "
                                      f"1 stratum x {cfg.K_s} clusters; one hamsm_assign pass over the resident child frames per step",
                          "frames": n, "repeats_of_the_1M_frames": reps},
               "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                            "traffic": pmc_kernel_bytes("k_assign_filter", "c2" if reps == 1 else f"c2_x{reps}"),
                            "traffic_source": (f"profiles/{os.path.basename(pmc_file('c2' if reps == 1 else f'c2_x{reps}'))} (committed "
                                               "rocprofv3 --pmc passes of this workload; not measured in this run)")
                            if pmc_file("c2" if reps == 1 else f"c2_x{reps}") else None,
                            "kernel": "k_assign_filter (bf16 x 3 ranking pass on the matrix unit) + k_assign_sorted (exact f64 pass over "
                                      "the frames the ranking cannot certify)",
                            "algorithmic_bytes_per_launch": bytes_alg, "flop_per_launch": flops, "avg_launch_ms": k_ms,
                            "achieved_TFLOPs": tf, "mfma_roof_TFLOPs": PEAK_BF16_MFMA_TFLOPS / 3.0,
                            "frac_of_mfma_roof": tf / (PEAK_BF16_MFMA_TFLOPS / 3.0),
                            "which_roof": "52 B and 4000 flop per frame = 77 flop/B, under the ridge of the bf16/3 ranking pass "
                                          "(104 flop/B): HBM binds"},
               "stage_ms_per_step": {"assign_kernels": k_ms, "bucket_passes": o_ms, "host_and_gaps": elapsed * 1e3 / steps - k_ms - o_ms},
               "whole_call": {"GBps": bytes_alg / (elapsed / steps) / 1e9, "frac": bytes_alg / (elapsed / steps) / 1e9 / PEAK_HBM_GBS},
               "step_incl_host_labels": {"ms_per_step": step_labels_s * 1e3, "value": n / step_labels_s, "unit": "frames/s"},
               "label_checksum": checksum}
        if cpu:
            import subprocess

            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--c2-cpu-worker", str(n1)], capture_output=True, text=True,
                               check=True, env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1"))
            cb = json.loads(r.stdout.strip().splitlines()[-1])
            if cb.pop("label_checksum") != checksum:
                raise SystemExit("[bench] c2: the GPU's labels differ from the CPU oracle's on the same frames")
            cb["labels_equal_gpu"] = True
            cb["gpu_over_cpu"] = out["value"] / cb["value"]
            out["cpu_baseline"] = cb
        log(f"c2 x{reps}: {out['ms_per_step']:.3f} ms per step, kernels {k_ms:.3f} ms = {gbs:.0f} GB/s algorithmic")
        return out
    finally:
        h.close()


def pmc_file(workload):
    """the committed rocprofv3 --pmc summary of this workload (latest round first); None when there is none"""
    names = {"c3": ["r05_pmc_summary.csv", "r04_pmc_summary.csv", "r03_pmc_summary.csv", "r02_pmc_summary.csv"],
             "c5s": ["r05_pmc_summary_c5s.csv", "r04_pmc_summary_c5s.csv"], "c2": ["r05_pmc_summary_c2.csv"],
             "c2_x64": ["r05_pmc_summary_c2_x64.csv"],
             "c5": ["r05_pmc_summary_c5s.csv", "r04_pmc_summary_c5s.csv"]}.get(workload, [])   # (c5's launches are c5s's: same per-iteration shapes)
    for n in names:
        p = os.path.join(ROOT, "profiles", n)
        if os.path.exists(p):
            return p
    return None


def pmc_kernel_bytes(name_part, workload="c3"):
    """(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch of the kernel whose name contains `name_part`, from the committed
    PMC summary of the workload; None when absent"""
    path = pmc_file(workload)
    if path is None:
        return None
    v = {}
    for line in open(path).read().splitlines()[1:]:
        k, c, _, x = line.rsplit(",", 3)
        if name_part in k:
            v[c] = float(x)
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        return None
    return (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0


def step_roofline(cfg, n_iters, n_seg, e_ms, e_n, m_ms, e_events_ms, ranked=False):
    """`roofline` of the bench line = the DOMINANT kernel of the timed step: the E-step of a training step -- `k_step_assign`
    (f64 matrix unit; C3: 48 % of all kernel time, with its M-step 78 %) or, for calls that are bound by that unit (>= 1 GFLOP per
    launch group: BASELINE config 5's shape), `k_step_assign_ranked` (bf16 matrix unit, exact f64 where the ranking cannot decide).
    One launch serves the batches of the two strata that share a queue: algorithmic flops = 2 strata x n_b x K_s x D x 2 (SURVEY
    §8d: 2 K_s D per row), n_b = the rows of one WE iteration that fall into one stratum outside basis / target; duration = every
    128th launch on the library's queue, in this run: by the device's 100 MHz clock from the first wave's entry to the last wave's
    exit (`avg_launch_ms`, what rocprofv3's kernel trace reports), and event to event (`avg_launch_ms_hip_events`: the same launch
    INCLUDING its kernel boundary, ~3 us more -- an event pair cannot bracket a 10 us kernel any tighter)."""
    S, K_s, D = cfg.S, cfg.K_s, cfg.D
    strata_per_launch = 2 if S >= 2 else 1
    n_b = cfg.segs_per_iter * (1.0 - 0.1 / S) / S
    flops = strata_per_launch * n_b * 2.0 * K_s * D
    bytes_alg = strata_per_launch * n_b * (D * 4 + 4)         # rows read + labels; centroid fragments are L2-resident
    e_ms = e_ms or float("nan")
    tf = flops / (e_ms * 1e-3) / 1e12
    kname = "k_step_assign_ranked" if ranked else "k_step_assign<"
    traffic = pmc_kernel_bytes(kname, cfg.name)
    src = pmc_file(cfg.name)
    common = {"traffic": traffic,
              "traffic_source": (f"profiles/{os.path.basename(src)} (committed rocprofv3 --pmc passes of this workload's launches; "
                                 "not measured in this run)") if traffic is not None else None,
              "avg_launch_ms": e_ms, "avg_launch_ms_hip_events": e_events_ms, "launches_sampled": e_n, "flop_per_launch": flops,
              "algorithmic_bytes_per_launch": bytes_alg, "m_step_avg_launch_ms_hip_events": m_ms}
    if ranked:
        # every f32 product is formed from 3 of the 9 bf16 piece products (operands = exact sums of bf16 pieces): the roof of an
        # algorithmic flop on this path is the bf16 MFMA peak / 3, as for the assignment's light ranking form
        peak = PEAK_BF16_MFMA_TFLOPS / 3.0
        return {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, **common,
                "frac_of_f64_mfma_peak": tf / PEAK_F64_MFMA_TFLOPS,
                "kernel": "k_step_assign_ranked (v_mfma_f32_32x32x16_bf16 on fragments the M-step maintains; rows the ranking cannot "
                          "certify: two f64 fma chains, or v_mfma_f64_16x16x4_f64 over all centroids), the E-step of one training step "
                          "for the two strata of a queue",
                "note": "what binds it is the fragment stream out of L2 (every workgroup reads its stratum's 256 B per centroid) and the "
                        "chain prologue -> sweep -> merge -> settle of 195 workgroups per launch, not the matrix unit (DESIGN.md 5.3e)"}
    # what binds this kernel at C3 is the latency of its dependent chain, not the matrix unit; `frac` stays
    # the fraction of the f64-MFMA peak (the roof its arithmetic would meet), `bound` says what it actually hits
    return {"bound": "latency", "roof": "mfma", "achieved": tf, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F64_MFMA_TFLOPS, **common,
            "kernel": "k_step_assign (v_mfma_f64_16x16x4_f64), the E-step of one training step for the two strata of a queue",
            "note": "latency-bound: ~2 500-row batches, 2 x n_iterations dependent launches per queue (DESIGN.md 5.3, 5.3a); the "
                    "assignment kernels have their own object, roofline_assign"}


def roofline_by_kernel(cfg, n_iters, n_seg, timers, steps, assign_tflops, assign_peak, assign_avg_ms):
    """One entry per stage of the timed step, from the HIP-event stage timers of THIS run (the library brackets every launch
    group with events on its stream): device milliseconds per step, the algorithmic work of SURVEY.md §8d, and the fraction
    of the roof that binds the stage.  The training steps dominate the step and are latency-bound."""
    per = {k: v / steps * 1e3 for k, v in timers.items()}
    S, K_s, D = cfg.S, cfg.K_s, cfg.D
    rows_trained = n_seg * (1.0 - 0.1 / S)            # parent pcoord outside basis (< 0.05) and target (> S - 0.05)
    train_flops = 2.0 * rows_trained * K_s * D        # E-step distances (the M-step adds 2 D per row)
    train_bytes = rows_trained * (2 * (D * 4 + 8) + 8 * D)   # rows read by the E-step, f64 copy written and re-read by the M-step
    out = [
        {"stage": "training steps", "kernels": "k_step_assign (v_mfma_f64_16x16x4_f64; calls of >= 1 GFLOP per launch group: "
                                               "k_step_assign_ranked, bf16 matrix unit) + k_step_update, one pair per WE iteration "
                                               "and queue, two strata per launch", "launches_per_step": 4 * (n_iters - 0),
         "device_ms_per_step": per["step"], "flop": train_flops, "achieved_TFLOPs": train_flops / (per["step"] * 1e-3) / 1e12,
         "peak_TFLOPs": PEAK_F64_MFMA_TFLOPS, "frac": train_flops / (per["step"] * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS,
         "algorithmic_GBps": train_bytes / (per["step"] * 1e-3) / 1e9, "bound": "latency: a chain of 2 x n_iterations dependent "
                                                                                 "launches of ~8 us on ~2500-row batches"},
        {"stage": "assign (2 launches)", "kernels": "k_assign_filter + k_assign_sorted", "device_ms_per_step": per["assign"],
         "avg_launch_ms": assign_avg_ms, "achieved_TFLOPs": assign_tflops, "peak_TFLOPs": assign_peak, "frac": assign_tflops / assign_peak,
         "bound": "mfma / vector issue"},
        {"stage": "k-means++ seeding", "kernels": "k_seed_gather + k_seed_gram + k_seed_g", "device_ms_per_step": per["seed"],
         "bound": "latency: K_s - 1 sequential greedy steps per stratum"},
        {"stage": "flux matrix", "kernels": "k_flux_xcd + k_flux_fold (atomic mode) or rocPRIM sort + 2 ordered passes",
         "device_ms_per_step": per["flux"], "bytes": 32.0 * n_seg, "achieved_GBps": 32.0 * n_seg / (per["flux"] * 1e-3) / 1e9
         if per["flux"] > 0 else None, "peak_GBps": PEAK_HBM_GBS, "frac": 32.0 * n_seg / (per["flux"] * 1e-3) / 1e9 / PEAK_HBM_GBS
         if per["flux"] > 0 else None, "bound": "L2 atomics"},
        {"stage": "bucket passes + training prepass", "kernels": "k_sort_count / k_sort_offsets / k_sort_fill, k_prepass, k_build_rows_wave",
         "device_ms_per_step": per["other"] + per["prepass"], "bytes": (3 * 12.0 * 2 + 12.0) * n_seg, "bound": "hbm"},
    ]
    return out


def ingest_leg(h, cfg, n_it, log):
    """f2: `stage_iterations` -- two page-locked slabs, the next block filled while the current one is uploaded on the copy stream
    -- from per-iteration `.npz` files (stored members = byte ranges of the files) and from the run-level raw layout
    (`NpyDirectorySource`), both read file -> slab by the library's threads (`hamsm_host_read_parts`), and from iterations held in
    memory (`hamsm_host_copy_parts`).  The sources are fresh objects: the `.npz` files' headers are parsed inside the figure.  Everything is inside the figure: slab allocation and release, device allocations, the NaN scan."""
    import shutil
    import tempfile
    from msm_we_amd.data import InMemorySource, NpyDirectorySource, NpzDirectorySource

    lat = synthetic.latent_centres(cfg)
    its = {it: synthetic.make_iteration(cfg, it, lat) for it in range(1, n_it + 2)}
    tmp = tempfile.mkdtemp(prefix="hamsm_ingest_")
    try:
        src = NpzDirectorySource.export(InMemorySource(its), os.path.join(tmp, "npz"))
        raw = NpyDirectorySource.export(InMemorySource(its), os.path.join(tmp, "npy"))
        res = {}
        for name, source in (("npz_files", NpzDirectorySource(src.dir)), ("raw_files", raw), ("in_memory", InMemorySource(its))):
            m = modelWE(h=h)
            m.pcoord_ndim = 1
            m.set_iteration_source(source)
            m.get_iterations()
            t0 = time.perf_counter()
            st = m.stage_iterations()
            h.sync()
            dt = time.perf_counter() - t0
            # the figure is the throughput of a VERIFIED staging -- every staged array is read back and must
            # equal the source byte for byte (a slab rewritten before its copy drained, a block landing at the wrong offset or a
            # short read would show here); the run stops otherwise
            keep = [i for i in sorted(its) if i <= m.maxIter]
            for key, buf in (("Xp", st.Xp), ("Xc", st.Xc), ("p0", st.p0), ("p1", st.p1), ("w", st.w)):
                want = np.concatenate([np.asarray(its[i][key]).reshape(len(its[i]["w"]), -1) for i in keep])
                got = buf.to_host().reshape(-1)[: want.size].reshape(want.shape)
                if got.dtype != want.dtype or not np.array_equal(got.view(np.uint8), want.view(np.uint8)):
                    bad = np.nonzero((got != want).any(axis=1))[0]
                    raise SystemExit(f"[bench] ingest leg {name}: staged {key} differs from the source in {len(bad)} rows "
                                     f"(first: {bad[:5].tolist()}) -- the staging pipeline corrupted its data")
                del want, got
            res[name] = {"seconds": dt, "GBps": m.ingest_stats["bytes"] / dt / 1e9, "bytes": m.ingest_stats["bytes"], "verified": "every staged array read back == source, byte for byte",
                         "blocks": m.ingest_stats["blocks"], "frames": int(st.N), "readers": m.ingest_stats.get("readers"), "direct_read_blocks": m.ingest_stats.get("direct_read_blocks"),
                         "host_waits_s": m.ingest_stats.get("host_waits_s")}
            for b in (st.Xp, st.Xc, st.p0, st.p1, st.w):
                b.free()
        log(f"ingest: {res}")
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic(cfg, n_iters):
    """HBM bytes per assign launch (ranking pass + exact pass) from the committed rocprofv3 PMC passes of this same workload
    (separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, calibrated
    here on k_sort_count's 80 MB pcoord read)."""
    if n_iters != cfg.n_iters or cfg.name not in ("c3", "c5s"):
        return None
    filt = os.environ.get("HAMSM_ASSIGN_FILTER", "1") != "0"
    a = pmc_kernel_bytes("k_assign_sorted", cfg.name)
    b = pmc_kernel_bytes("k_assign_filter", cfg.name) if filt else 0.0
    return None if a is None or b is None else a + b


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--c2-cpu-worker":
        return c2_cpu_worker(int(sys.argv[2]))
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_baseline_worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c5", "c5s", "tiny"],
                    help="c3 = BASELINE configs[2] (the configuration the metric is quoted on; default); c5 = configs[4] at its "
                         "stated size on ONE GPU (50 M frames x 64 features, 4 x 1250 clusters + the ~5000-state solve; 25.6 GB of "
                         "frames, staged in slabs); c5s = a 10 M-frame slice of c5")
    ap.add_argument("--iters", type=int, default=0, help="override iterations per rank (debug)")
    ap.add_argument("--flux-mode", type=int, default=1, help="1 = deterministic sort-based (the product default), 0 = f64 atomics")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=0, help="iterations of the CPU baseline's sample (0: about the work of 128 C3 iterations, ~10 s on one core)")
    ap.add_argument("--cpu-cores", type=int, default=min(16, os.cpu_count() or 1), help="processes of the all-cores CPU leg")
    ap.add_argument("--no-full-pipeline", action="store_true")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1.  strong (default; BASELINE config 4: the workload's 10 M frames in all, split over the ranks) or "
                         "weak (every rank its own full shard: N x the frames)")
    ap.add_argument("--train-on", default=None, choices=["rank0", "all", "strata"],
                    help="sharded runs.  all: rank 0 trains on every rank's iterations (gathered to it at staging): N ranks == one "
                         "process on the same data.  strata: the same model, every rank receives every shard's training frames and "
                         "trains the strata s %% N == rank, stratum s's model travels from its rank.  Default for strong scaling: "
                         "strata where a stratum's training call is matrix-bound (>= 0.5 GFLOP: BASELINE config 5's shape, 2.2x "
                         "shorter pass with a stratum per GPU), all where it is latency-bound (config 3 / 4: one stratum's chain "
                         "ALONE takes 24.8 us per call, the four chains overlapped on one chip 17.4 -- tools/onestratum.py).  "
                         "rank0 (default for weak scaling): rank 0 trains on its own block of iterations only")
    ap.add_argument("--no-ingest", action="store_true")
    ap.add_argument("--secondary", default=None,
                    help="N = 1: comma-separated workloads measured after the primary one and reported under `secondary` in the same "
                         "line (c2 = BASELINE configs[1], c2_x64 = the same frames resident 64 times, or any --workload name).  "
                         "Default for the default workload (c3): c2,c2_x64,c5 -- every single-GPU BASELINE config in one "
                         "driver-timed record; for any other workload: none")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--c2-frames", type=int, default=0, help="(debug / CPU rehearsal) frames of config 2 instead of its 1 000 000")
    ap.add_argument("--lib", default=None, help="(kernel A/B experiments) path of an alternative build of the HIP library")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist

        # torch only BOOTSTRAPS the run -- the 128-byte RCCL id of the library's own communicator, two scalars, the barriers --
        # over gloo on CPU tensors: it never initialises a GPU (the collectives of the product are the library's, csrc/comm.hip)
        dev = "cpu"
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"

    seed = 1337
    exchange = "none (one rank)"

    # (HAMSM_BENCH_DEVICE: rehearsals of the N > 1 flow on a box with fewer GPUs than ranks)
    device = int(os.environ.get("HAMSM_BENCH_DEVICE", local_rank))
    h = _abi.Hamsm(lib=ctypes.CDLL(args.lib) if args.lib else None, device=device)  # the HIP library; raises if not built
    assert args.lib or h.backend == "hip-gfx950"
    if world > 1:
        uid = [h.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        try:
            h.comm_init(world, rank, uid[0])   # RCCL communicator inside the library (csrc/comm.hip)
            ok = 1
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: in-library RCCL init failed ({e}); exchange steps go through torch.distributed",
                  file=sys.stderr, flush=True)
            ok = 0
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        exchange = "rccl-in-library"
        if int(flag.item()) == 0:
            # the collectives of the product are the library's RCCL calls (csrc/comm.hip).  A run that cannot create that
            # communicator is NOT a measurement of them: it stops, unless a rehearsal asks for the host-staged stand-in
            if os.environ.get("HAMSM_BENCH_ALLOW_FALLBACK", "0") != "1":
                raise SystemExit(f"[bench] rank {rank}: in-library RCCL communicator unavailable on some rank; refusing to fall back "
                                 "to host-staged torch.distributed collectives (set HAMSM_BENCH_ALLOW_FALLBACK=1 to rehearse)")
            if ok:
                h.comm_destroy()
            torch_exchange(h, dist, torch, rank, world, dev)
            exchange = "torch-fallback (host-staged; not the product path)"

    def measure(h, workload, steps, warmup, ingest, cpu_iters_arg):
        """one workload's bench object: stage the shard, W untimed + K timed steps between barriers, then the legs that are not
        part of `value` (labels to the host, both flux modes, ingest, the full build_analyze_model sequence, the CPU baseline)"""
        cfg = bench_configs()[workload]
        n_total = args.iters or cfg.n_iters
        if args.scaling == "strong" and world > 1:
            per = n_total // world
            first_iter = 1 + rank * per
            n_iters = per if rank < world - 1 else n_total - per * (world - 1)
        else:
            n_iters, first_iter = n_total, 1 + rank * n_total
        call_flop = 2.0 * (cfg.segs_per_iter / cfg.S) * cfg.K_s * cfg.D   # one stratum's training call
        train_on = args.train_on or (("strata" if call_flop >= 5e8 else "all") if args.scaling == "strong" else "rank0")
        h.timers(reset=True)
        log(f"workload {cfg.name}: {n_iters} iterations x {cfg.segs_per_iter} segments x {cfg.D} features, "
            f"{cfg.S} strata x {cfg.K_s} clusters, rank shard starts at iteration {first_iter}; scaling {args.scaling}, "
            f"training on {train_on}, exchange {exchange}")
        if n_iters * cfg.segs_per_iter * cfg.D * 8 > (8 << 30) or os.environ.get("HAMSM_BENCH_STREAM") == "1":
            store, offs = synthetic.stage_streamed(h, cfg, first_iter, n_iters, log)   # (c5: never whole on the host)
        else:
            Xp, Xc, p0, p1, w, offs = make_shard(cfg, first_iter, n_iters, log)
            store = DeviceStore.from_host(h, Xp, Xc, p0, p1, w, offs)
            del Xp, Xc
        n_seg = int(offs[-1])
        m = new_model(h, cfg, store, n_iters)
        m.train_shards = train_on
        n_seg_all = n_seg
        if world > 1:
            t = torch.tensor([n_seg], device=dev, dtype=torch.int64)
            dist.all_reduce(t)
            n_seg_all = int(t.item())
            m.training_store()   # (train-on all: the one-off gather of the training frames to rank 0 belongs to staging)

        def barrier():
            h.sync()
            if dist is not None:
                dist.barrier()   # (h.sync() above drained this rank's GPU queue: the library's stream is the only one in use)

        for _ in range(warmup):
            one_step(m, cfg, seed, args.flux_mode)
        barrier()
        h.timers(reset=True)
        if hasattr(h, "step_kernel_ms"):
            h.step_kernel_ms(0, reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step(m, cfg, seed, args.flux_mode)
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        timers = h.timers()   # HIP events on the library's stream around every kernel (lazy: no sync on the launch path)
        # training E-step on the bf16 matrix unit (train.hip, k_step_assign_ranked; taken by size): rows certified by the ranking /
        # settled between two candidates by two f64 chains / swept exactly, over the timed steps
        rank_stats = None
        if hasattr(h.lib, "hamsm_debug_rank_stats"):
            try:
                out4 = (ctypes.c_ulonglong * 4)()
                h.lib.hamsm_debug_rank_stats.restype = ctypes.c_int
                h.lib.hamsm_debug_rank_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
                if h.lib.hamsm_debug_rank_stats(h.ctx, out4, 0, -1) == 0 and out4[3] > 0:
                    tot = max(1, out4[0] + out4[1] + out4[2])
                    rank_stats = {"runs": int(out4[3]), "rows_certified": int(out4[0]), "rows_two_candidates": int(out4[1]),
                                  "rows_swept_exactly": int(out4[2]), "frac_certified": out4[0] / tot, "frac_two": out4[1] / tot,
                                  "frac_swept": out4[2] / tot}
            except (AttributeError, OSError):
                rank_stats = None
        e_ms, e_n = h.step_kernel_ms(0)   # the dominant kernel, live: events around every 128th k_step_assign / k_step_update launch
        m_ms, m_n = h.step_kernel_ms(1)
        e_dev_ms, _ = h.step_kernel_ms(2)  # ... and by the device clock (first wave in -> last wave out, no launch boundary)
        # what the reference's post-condition adds when the caller wants host arrays (not part of the timed step: `dtrajs` /
        # `pair_dtrajs` are materialised on first access): 2 x N int32 labels D2H + the per-iteration views
        # (`_abi.HostBlock`: one device-to-host copy of 8 N bytes into a page-locked block of the process; the first call of a size
        # also page-locks the block, ~0.3 ms per MB -- later calls reuse it)
        t0 = time.perf_counter()
        _ = m.dtrajs
        labels_to_host_first_ms = (time.perf_counter() - t0) * 1e3
        del _
        m._dtrajs_cache = None
        t0 = time.perf_counter()
        _ = m.dtrajs
        labels_to_host_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        _ = m.pair_dtrajs
        pairs_to_host_ms = (time.perf_counter() - t0) * 1e3
        del _
        # the step as a caller who reads `model.dtrajs` after every build sees it
        n_l = min(steps, 5)
        for _ in range(min(warmup, 2)):   # (untimed, like the headline's warm-up: the page-locked blocks the arrays are views of are
            one_step(m, cfg, seed, args.flux_mode)   # allocated once per size and then alternate -- a caller keeps the previous list
            _d = m.dtrajs                            # until the next one exists)
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_l):
            one_step(m, cfg, seed, args.flux_mode)
            _d = m.dtrajs
        barrier()
        step_labels_s = (time.perf_counter() - t0) / n_l
        del _d
        # the flux build in both modes (the timed step uses --flux-mode; the product default is the deterministic one)
        flux_ms = {}
        for mode, name in ((0, "atomic"), (1, "deterministic")):
            m.flux_mode = mode
            m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter)
            h.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter)
            h.sync()
            flux_ms[name] = (time.perf_counter() - t0) * 1e3 / 3
        m.flux_mode = args.flux_mode
        assign_ms = h.last_kernel_ms(0)
        n_assign_launch = 2 * steps
        assign_avg_ms = timers["assign"] * 1e3 / n_assign_launch   # the distance kernels alone (f32 ranking pass + exact f64 pass over
                                                                   # the undecided frames); the bucket passes are "other"
        rows_per_launch = n_seg                                   # one launch = all parent (or child) frames
        flops = 2.0 * cfg.K_s * cfg.D * rows_per_launch           # 2*K_s*D per frame
        bytes_alg = (cfg.D * 4 + 8 + 4) * rows_per_launch         # D*s + 8*ndim + 4 per frame
        assign_avg_ms = assign_avg_ms or float("nan")  # (the CPU rehearsal has no kernel timers)
        tflops = flops / (assign_avg_ms * 1e-3) / 1e12
        K = cfg.K_s * cfg.S
        extra = {}
        # which form of the ranking pass the last assign ran (library hook; the CPU rehearsal has none): 2 = bf16 x 3 with 6 piece
        # products, 3 = the light bf16 form with 3
        form = None
        if hasattr(h.lib, "hamsm_debug_filter_form"):
            try:
                h.lib.hamsm_debug_filter_form.restype = ctypes.c_int
                h.lib.hamsm_debug_filter_form.argtypes = [ctypes.c_void_p]
                form = int(h.lib.hamsm_debug_filter_form(h.ctx))
            except (AttributeError, OSError):
                form = None
        if form in (2, 3) or (form is None and os.environ.get("HAMSM_ASSIGN_FILTER", "1") != "0"
                              and os.environ.get("HAMSM_FILTER_BF16", "1") != "0" and cfg.D <= 32):
            # every f32 product is formed from bf16 piece products (operands = exact sums of three bf16 pieces): 6 of the 9 in the full
            # form, 3 in the light form that goes first (wider bound, more frames left to the exact kernel): the roof of an algorithmic
            # flop on this path is the bf16 MFMA peak / products
            products = 3.0 if form == 3 else 6.0
            peak = PEAK_BF16_MFMA_TFLOPS / products
            blocks = (cfg.D + 16) // 16
            executed = flops * (products * 16 * blocks / cfg.D) * (((cfg.K_s + 31) // 32) * 32 / cfg.K_s)
            extra = {"bf16_mfma_peak": PEAK_BF16_MFMA_TFLOPS, "executed_bf16_TFLOPs": executed / (assign_avg_ms * 1e-3) / 1e12,
                     "frac_of_f32_mfma_peak": tflops / PEAK_F32_MFMA_TFLOPS, "bf16_products_per_f32_product": products}
            kernel = (f"k_assign_filter (v_mfma_f32_32x32x16_bf16, f32 operands as three exact bf16 pieces, {int(products)} piece products) "
                      "+ k_assign_sorted (v_mfma_f64_16x16x4_f64) on the undecided frames")
            arithmetic = ("centroids ranked on the bf16 matrix unit; a label is final only if the gap exceeds a proven bound on the "
                          "difference to the f64 contract (dropped piece products included), otherwise the exact f64 kernel decides: "
                          "labels identical to the f64 contract")
        elif os.environ.get("HAMSM_ASSIGN_FILTER", "1") != "0":
            peak = PEAK_F32_MFMA_TFLOPS
            kernel = "k_assign_filter (v_mfma_f32_32x32x2_f32) + k_assign_sorted (v_mfma_f64_16x16x4_f64) on the undecided frames"
            arithmetic = ("centroids ranked in f32 on the matrix unit; a label is final only if the f32 gap exceeds a proven bound "
                          "on the f32-vs-f64 difference, otherwise the exact f64 kernel decides: labels identical to the f64 contract")
        else:
            peak = PEAK_F64_MFMA_TFLOPS
            kernel = "k_assign_sorted (v_mfma_f64_16x16x4_f64)"
            arithmetic = "f64 in-order fma chains on the matrix unit"
        out = {
            "metric": "frames/sec clustered + fluxmatrix-built",
            "value": n_seg_all * steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed * 1e3 / steps,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "n/a",
            "vs_baseline": None,
            "dtype": "f64",
            "dtype_detail": "training, flux, statistics and the exact assignment kernel compute in f64; the assignment's ranking pass "
                            "computes in bf16 x 3 (f32-equivalent) or f32 and only finalises labels its error bound certifies against f64",
            "data": "synthetic",
            "config": {"workload": f"{cfg.name}: {n_iters} WE iterations x {cfg.segs_per_iter} segments per GPU, "
                                   f"{cfg.D} f32 features, {cfg.S} strata x {cfg.K_s} clusters (K={K}); "
                                   f"train + assign(2N) + flux per step", "frames_per_gpu": n_seg,
                       "flux_mode": "atomic" if args.flux_mode == 0 else "deterministic",
                       "frames_total": n_seg_all, "parallelism": f"iteration-sharded x{world}",
                       "training": ({"all": "rank 0 trains on every rank's iterations (gathered to it at staging): same model as one process "
                                            "on all the data",
                                     "strata": "every rank's iterations reach every rank at staging; stratum s trains on rank s % N and its "
                                               "model travels from there: same model as one process on all the data, training spread over "
                                               "min(N, S) GPUs",
                                     "rank0": "rank 0 trains on its own block of iterations only (iters_to_use); --train-on all / strata "
                                              "use every shard"}[train_on])
                       if world > 1 else "all resident iterations", "exchange": exchange},
            "roofline": step_roofline(cfg, n_iters, n_seg, e_dev_ms or e_ms, e_n, m_ms, e_ms, ranked=rank_stats is not None),
            "roofline_assign": {"bound": "mfma", "achieved": tflops, "peak": peak, "unit": "TFLOP/s",
                                "frac": tflops / peak, "traffic": pmc_traffic(cfg, n_iters),
                                "traffic_source": (f"profiles/{os.path.basename(pmc_file(cfg.name))} (committed rocprofv3 --pmc passes of this "
                                                   "workload); not measured in this run") if pmc_traffic(cfg, n_iters) is not None else None,
                                "kernel": kernel, "arithmetic": arithmetic,
                                "f64_mfma_peak": PEAK_F64_MFMA_TFLOPS, "frac_of_f64_peak": tflops / PEAK_F64_MFMA_TFLOPS, **extra,
                                "avg_launch_ms": assign_avg_ms, "last_launch_ms": assign_ms,
                                "algorithmic_GBps": bytes_alg / (assign_avg_ms * 1e-3) / 1e9,
                                "hbm_frac": bytes_alg / (assign_avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
            "stage_seconds_per_step": {k: v / steps for k, v in timers.items()},
            "roofline_by_kernel": roofline_by_kernel(cfg, n_iters, n_seg, timers, steps, tflops, peak, assign_avg_ms),
            "labels_to_host_ms": labels_to_host_ms,
            "labels_to_host": {"dtrajs_ms": labels_to_host_ms, "dtrajs_first_call_ms": labels_to_host_first_ms,
                               "pair_dtrajs_ms": pairs_to_host_ms, "bytes_dtrajs": 8 * n_seg, "bytes_pair_dtrajs": 16 * n_seg,
                               "how": "int64 labels widened (and interleaved) on the device, one copy each into a page-locked "
                                      "process-level block; per-iteration arrays are views; the first call of a size also page-locks"},
            "step_incl_host_labels": {"ms_per_step": step_labels_s * 1e3, "value": n_seg_all / step_labels_s, "unit": "frames/s",
                                      "steps": n_l, "what": "the same step followed by `model.dtrajs` (the reference's host-side "
                                                            "post-condition, _clustering.py:1206-1240) every time"},
            "flux_ms": flux_ms,
            "training_estep": ("ranked on the bf16 matrix unit (k_step_assign_ranked), exact f64 where the ranking cannot decide"
                               if rank_stats else "f64 matrix unit (k_step_assign)"),
            "training_estep_rows": rank_stats,
        }
        if world == 1 and ingest:
            # (a bounded sample: ~1 GB of frames, 400 iterations at C3; HAMSM_BENCH_INGEST_ITERS: another number of iterations.  The
            # fixed costs -- page-locking and releasing two upload slabs, the device allocations: ~20 ms -- are inside the figure)
            n_ing = int(os.environ.get("HAMSM_BENCH_INGEST_ITERS", 0)) or min(400, max(4, int(1.056e9 / (cfg.segs_per_iter * cfg.D * 8))))
            out["ingest"] = ingest_leg(h, cfg, min(n_iters, n_ing), log)
        if not args.no_full_pipeline and (world == 1 or os.environ.get("HAMSM_BENCH_FULL_PIPELINE_SHARDED", "1") != "0"):
            # BASELINE.json's second figure (steady-state solve wall-time) and SURVEY §8d's build_analyze_model-equivalent.
            # N > 1: the same sequence on the shards -- its 4 assign passes and 6 flux builds scale with the
            # ranks, training does not; block validation splits every rank's own iterations into the blocks (the groups'
            # matrices are all-reduced like the full one); every rank solves the same all-reduced matrix
            log("timing the full build_analyze_model sequence (clean, T, solve, block validation) ...")
            barrier()
            fp = full_pipeline(m, cfg, seed, args.flux_mode)
            barrier()
            if dist is not None:
                t = torch.tensor([fp["total_seconds"]], device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                fp["total_seconds"] = float(t.item())
            fp["frames_per_s"] = n_seg_all / fp["total_seconds"]
            out["full_pipeline"] = fp
            out["solve"] = {"states": fp["states"], "tmatrix_csr_s": fp["seconds"]["tmatrix_csr"],
                            "steady_state_s": fp["seconds"]["steady_state"], "where": "T built as CSR on the GPU; scipy solve on host"}
        if rank == 0 and not args.no_cpu_baseline and world == 1:
            log(f"timing the CPU oracle on a bounded sample of {workload} ...")
            # a BOUNDED sample: the work of 128 C3 iterations (~10 s on one core), never fewer than 2 iterations
            cpu_iters = cpu_iters_arg or max(2, int(128 * (10000 * 250 * 30) / (cfg.segs_per_iter * cfg.K_s * cfg.D)))
            out["cpu_baseline"] = cpu_baseline(workload, min(cpu_iters, n_iters), seed, args.cpu_cores)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        return out

    if args.workload == "c2":
        if world > 1:
            raise SystemExit("--workload c2 is a single-GPU workload (BASELINE configs[1]: 1 x MI355X distance / assign only)")
        h.close()
        out = measure_c2(device, args, log, int(os.environ.get("HAMSM_BENCH_C2_REPS", "1")), cpu=not args.no_cpu_baseline)
        out.update({"scaling": "n/a", "vs_baseline": None})
    else:
        out = measure(h, args.workload, args.steps, args.warmup, not args.no_ingest, args.cpu_iters)
    secondary = args.secondary if args.secondary is not None else ("c2,c2_x64,c5" if args.workload == "c3" and not args.iters else "")
    if world == 1 and secondary and not args.no_secondary:
        # every single-GPU BASELINE config as a driver-timed record of the SAME invocation.  `value` stays
        # config 3's (the configuration the metric is quoted on); config 2 (distance / assign only, at its stated 1 M frames and
        # as a 64 M-frame repeat of the same shape: the kernel's asymptote) and config 5 at its stated size on this one GPU
        # follow under `secondary`, each with its own roofline and cpu_baseline.
        h.close()   # (everything the primary workload held in HBM goes with its context; the secondary workloads use their own)
        sec = {}
        for name in [x.strip() for x in secondary.split(",") if x.strip()]:
            log(f"secondary workload {name} ...")
            if name.startswith("c2"):
                reps = int(name.split("_x")[1]) if "_x" in name else 1
                sec[name] = measure_c2(device, args, log, reps, cpu=(reps == 1 and not args.no_cpu_baseline))
            else:
                hs = _abi.Hamsm(lib=ctypes.CDLL(args.lib) if args.lib else None, device=device)
                big = bench_configs()[name].n_frames > 20_000_000
                sec[name] = measure(hs, name, 3 if big else args.steps, 1 if big else args.warmup, False, 0)
                hs.close()
        out["secondary"] = sec
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        h.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

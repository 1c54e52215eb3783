"""N>1 path on CPU: world_size 2 over gloo.  The sharding logic of the mixins (rank 0 trains, model
broadcast, per-rank discretization, flux all-reduce) runs on the oracle backend; the two exchange
steps, which the product performs with RCCL inside the library (csrc/comm.hip), are served here by
torch.distributed/gloo on host copies -- a test-side stand-in for the collective only."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, RectilinearBinMapper, _build_oracle, golden_iterations, load_golden

torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_comm(h, dist, rank, world):
    """give a single-process ctx the collective entry points, implemented with gloo on host buffers"""
    h.nranks, h.rank = world, rank

    def allreduce(dtype, op=dist.ReduceOp.SUM):
        def f(d_buf, count):
            a = np.empty(count, dtype=dtype)
            h.d2h(a, d_buf)
            t = torch.from_numpy(a)
            dist.all_reduce(t, op=op)
            h.h2d(d_buf, t.numpy())
        return f

    h.allreduce_sum_f64 = allreduce(np.float64)
    h.allreduce_sum_i64 = allreduce(np.int64)
    h.allreduce_min_f64 = allreduce(np.float64, dist.ReduceOp.MIN)
    h.allreduce_max_f64 = allreduce(np.float64, dist.ReduceOp.MAX)

    def bcast(d_buf, nbytes, root=0):
        a = np.empty(nbytes, dtype=np.uint8)
        h.d2h(a, d_buf)
        t = torch.from_numpy(a)
        dist.broadcast(t, root)
        h.h2d(d_buf, t.numpy())

    h.bcast = bcast

    def bcast_model(root=0):
        state = [[h.get_centroids(s) for s in range(h.S)] if rank == root else None]
        dist.broadcast_object_list(state, src=root)
        if rank != root:
            for s, (Cm, counts, csq) in enumerate(state[0]):
                h.set_centroids(s, Cm if len(Cm) else None, csq=csq if len(Cm) else None, counts=counts if len(Cm) else None)

    h.bcast_model = bcast_model

    def bcast_strata(roots):
        for s in range(h.S):
            root = int(roots[s])
            state = [h.get_centroids(s) if rank == root else None]
            dist.broadcast_object_list(state, src=root)
            if rank != root:
                Cm, counts, csq = state[0]
                h.set_centroids(s, Cm if len(Cm) else None, csq=csq if len(Cm) else None, counts=counts if len(Cm) else None)

    h.bcast_strata = bcast_strata

    def gather_to_root(d_send, send_bytes, d_recv, recv_bytes, root=0):
        a = np.empty(int(send_bytes), dtype=np.uint8)
        h.d2h(a, d_send)
        parts = [None] * world if rank == root else None
        dist.gather_object(a, parts, dst=root)
        if rank == root:
            assert [len(p) for p in parts] == [int(b) for b in recv_bytes]
            h.h2d(d_recv, np.concatenate(parts))

    h.gather_to_root = gather_to_root

    def exchange(d_send, send_bytes, d_recv, recv_bytes):
        sb, rb = [int(x) for x in send_bytes], [int(x) for x in recv_bytes]
        a = np.empty(sum(sb), dtype=np.uint8)
        if a.size:
            h.d2h(a, d_send)
        cuts = np.cumsum([0] + sb)
        rows = [None] * world
        dist.all_gather_object(rows, [a[cuts[q]: cuts[q + 1]] for q in range(world)])
        mine = [rows[r][rank] for r in range(world)]
        assert [len(x) for x in mine] == rb, ([len(x) for x in mine], rb)    # what the library checks collectively
        if sum(rb):
            h.h2d(d_recv, np.concatenate(mine))

    h.exchange = exchange


def _shard_model(h, g, meta, first, last):
    from msm_we_amd.data import InMemorySource
    from msm_we_amd.model import modelWE

    its = golden_iterations(g)
    shard = {i - first + 1: its[i] for i in range(first, last + 2)}  # + the successor iteration
    m = modelWE(h=h)
    m.set_iteration_source(InMemorySource(shard))
    m.initialize(["g"], {"coords": None, "nAtoms": meta["D"], "coord_ndim": 1}, "g", basis_pcoord_bounds=meta["basis"],
                 target_pcoord_bounds=meta["target"], tau=meta["tau"], pcoord_ndim=1)
    m.get_iterations()
    m.dimReduce()
    return m


def _worker(rank, world, port, out, train_shards="all"):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from msm_we_amd import _abi

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, meta = load_golden("c1p_d10")
    h = _abi.Hamsm(lib=ctypes.CDLL(_build_oracle()))
    _gloo_comm(h, dist, rank, world)
    # loaded iterations 1..21 and 20..39: the reference's maxIter bookkeeping discretizes all but the last two loaded
    # ones (_data.py:860-866), i.e. 1..19 and 20..37 -- disjoint, and together what one process makes of 1..39
    m = _shard_model(h, g, meta, (1, 20)[rank], (20, 38)[rank])
    m.train_shards = train_shards
    bm = RectilinearBinMapper([np.array(meta["boundaries"], dtype=np.float32)])
    m.cluster_coordinates(meta["K_s"], streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm,
                          random_state=meta["seed"])
    m.get_fluxMatrix(0, first_iter=0, last_iter=m.maxIter)  # every local iteration with pairs
    res = dict(labels=np.concatenate(m.dtrajs), flux=m.fluxMatrixRaw, centers0=m.clusters.cluster_models[0].cluster_centers_,
               centers1=m.clusters.cluster_models[1].cluster_centers_, counts1=m.clusters.cluster_models[1]._counts,
               steps=[getattr(cm, "n_steps_", -1) for cm in m.clusters.cluster_models],
               remap=[m.clusters.we_remap[b] for b in range(2)])
    # the rest of build_analyze_model on the shards: cleaning (re-discretize + re-flux), cluster statistics, T, solve
    m.organize_fluxMatrix(use_ray=True)
    m.get_Tmatrix()
    m.get_steady_state()
    m.get_steady_state_target_flux()
    res.update(fluxMatrix=m.fluxMatrix, T=m.Tmatrix, pSS=m.pSS, J=m.JtargetSS, centers=m.targetRMSD_centers,
               minmax=m.targetRMSD_minmax, counts=m.cluster_pcoord_counts, labels2=np.concatenate(m.dtrajs))
    np.savez(out.format(rank), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["all", "strata"])
def test_two_ranks_equal_one_process_on_all_iterations(tmp_path, oracle_lib, mode):
    """`train_shards = "all"` (the default): the training frames of every shard are gathered to rank 0, which then trains
    on all iterations in order -- the two-rank run IS the single-process run of the same data: every label, the raw flux
    matrix, the cleaned model, T, the steady state.  `"strata"`: every rank receives every shard's training frames and trains
    the strata s % 2 == rank (independent chains, `_clustering.py:890-911`); stratum s's model travels from its rank
    (`hamsm_bcast_strata`) -- the same single-process model, bit for bit, with the training spread over the ranks."""
    import torch.multiprocessing as mp

    from msm_we_amd import _abi

    out = str(tmp_path / "rank{}.npz")
    mp.spawn(_worker, args=(2, _free_port(), out, mode), nprocs=2, join=True)
    r0, r1 = np.load(out.format(0)), np.load(out.format(1))
    g, meta = load_golden("c1p_d10")
    bm = RectilinearBinMapper([np.array(meta["boundaries"], dtype=np.float32)])
    h = _abi.Hamsm(lib=oracle_lib)
    c = _shard_model(h, g, meta, 1, 38)
    c.cluster_coordinates(meta["K_s"], streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=meta["seed"])
    labels = np.concatenate(c.dtrajs)
    assert np.array_equal(np.concatenate([r0["labels"], r1["labels"]]), labels)
    assert np.array_equal(r0["centers0"], c.clusters.cluster_models[0].cluster_centers_) and np.array_equal(r0["centers0"], r1["centers0"])
    assert np.array_equal(r0["centers1"], c.clusters.cluster_models[1].cluster_centers_) and np.array_equal(r0["centers1"], r1["centers1"])
    assert np.array_equal(r0["counts1"], c.clusters.cluster_models[1]._counts) and np.array_equal(r1["counts1"], r0["counts1"])
    if mode == "strata":   # the streaming bookkeeping of every stratum reached every rank
        assert list(r0["steps"]) == list(r1["steps"]) == [cm.n_steps_ for cm in c.clusters.cluster_models]
    c.get_fluxMatrix(0, first_iter=0, last_iter=c.maxIter)
    assert np.array_equal(r0["flux"], r1["flux"])
    np.testing.assert_allclose(r0["flux"], c.fluxMatrixRaw, rtol=1e-12, atol=1e-300)
    c.organize_fluxMatrix(use_ray=True)
    c.get_Tmatrix()
    c.get_steady_state()
    c.get_steady_state_target_flux()
    assert np.array_equal(np.concatenate([r0["labels2"], r1["labels2"]]), np.concatenate(c.dtrajs))
    for k in ("fluxMatrix", "T", "pSS", "J", "centers", "minmax", "counts"):
        assert np.array_equal(r0[k], r1[k], equal_nan=True), k
    np.testing.assert_allclose(r0["fluxMatrix"], c.fluxMatrix, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(r0["T"], c.Tmatrix, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(r0["pSS"], c.pSS, rtol=1e-8, atol=1e-300)
    assert abs(float(r0["J"]) - c.JtargetSS) <= 1e-8 * c.JtargetSS


def test_two_ranks_match_single_process(tmp_path, oracle_lib):
    """`train_shards = "rank0"`: rank 0 trains on its own block of iterations only (the reference's `iters_to_use`)"""
    import torch.multiprocessing as mp

    from msm_we_amd import _abi

    out = str(tmp_path / "rank{}.npz")
    mp.spawn(_worker, args=(2, _free_port(), out, "rank0"), nprocs=2, join=True)
    r0, r1 = np.load(out.format(0)), np.load(out.format(1))

    # single process: train on shard 0, reuse its centroids for shard 1, combine the flux by hand
    g, meta = load_golden("c1p_d10")
    bm = RectilinearBinMapper([np.array(meta["boundaries"], dtype=np.float32)])
    h = _abi.Hamsm(lib=oracle_lib)
    a = _shard_model(h, g, meta, 1, 20)
    a.cluster_coordinates(meta["K_s"], streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm,
                          random_state=meta["seed"])
    a.get_fluxMatrix(0, first_iter=0, last_iter=a.maxIter)
    la, Fa, na = np.concatenate(a.dtrajs), a.fluxMatrixRaw * (a.maxIter - 1), a.maxIter - 1
    b = _shard_model(h, g, meta, 20, 38)
    b.stage_iterations()
    b.clusters, b.n_clusters, b.clustering_method = a.clusters, a.n_clusters, "stratified"
    b.clusters.model = b
    b.launch_ray_discretization()
    b.get_fluxMatrix(0, first_iter=0, last_iter=b.maxIter)
    lb, Fb, nb = np.concatenate(b.dtrajs), b.fluxMatrixRaw * (b.maxIter - 1), b.maxIter - 1

    assert np.array_equal(r0["labels"], la) and np.array_equal(r1["labels"], lb)      # labels bit-exact
    assert np.array_equal(r0["centers0"], r1["centers0"])                              # broadcast model
    assert np.array_equal(r0["remap"], r1["remap"])
    assert np.array_equal(r0["flux"], r1["flux"])                                      # all ranks hold the same sum
    np.testing.assert_allclose(r0["flux"], (Fa + Fb) / (na + nb), rtol=1e-12, atol=1e-300)
    assert abs(r0["flux"].sum() - 1.0) < 1e-12

    # cleaned model: every rank ends with the same matrix, statistics and solution ...
    for k in ("fluxMatrix", "T", "pSS", "J", "centers", "minmax", "counts"):
        assert np.array_equal(r0[k], r1[k], equal_nan=True), k
    # ... and it is the single-process model of all 38 iterations trained on the first 19 (`iters_to_use`)
    c = _shard_model(h, g, meta, 1, 38)
    c.cluster_coordinates(meta["K_s"], streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm,
                          random_state=meta["seed"], iters_to_use=range(1, 20))
    c.get_fluxMatrix(0, first_iter=0, last_iter=c.maxIter)
    c.organize_fluxMatrix(use_ray=True)
    c.get_Tmatrix()
    c.get_steady_state()
    c.get_steady_state_target_flux()
    assert np.array_equal(np.concatenate([r0["labels2"], r1["labels2"]]), np.concatenate(c.dtrajs))
    assert np.array_equal(r0["counts"], c.cluster_pcoord_counts)
    np.testing.assert_allclose(r0["minmax"], c.targetRMSD_minmax, rtol=0, atol=0, equal_nan=True)
    np.testing.assert_allclose(r0["centers"], c.targetRMSD_centers, rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(r0["fluxMatrix"], c.fluxMatrix, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(r0["T"], c.Tmatrix, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(r0["pSS"], c.pSS, rtol=1e-8, atol=1e-300)
    assert abs(float(r0["J"]) - c.JtargetSS) <= 1e-8 * c.JtargetSS


def test_bench_two_rank_flow_on_cpu(tmp_path, oracle_lib):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one rank per "GPU"), rehearsed
    on the CPU: gloo process group, oracle library, and -- because the oracle has no collective backend -- the
    torch.distributed stand-in for the exchange steps.  Checks the contract line of rank 0."""
    import json
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS="1", HAMSM_BENCH_ALLOW_FALLBACK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--workload", "tiny", "--lib", _build_oracle()]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # ONE line, from rank 0
    out = json.loads(lines[0])
    # the default for N > 1 is BASELINE config 4's shape of run: the workload's frames split over the ranks (strong scaling), the
    # model of one process on the same data -- trained by rank 0 on everybody's iterations where a stratum's call is latency-bound
    # (this workload, C3 / C4), stratum by stratum over the ranks where it is matrix-bound (C5's shape)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["unit"] == "frames/s" and out["value"] > 0
    assert "cpu_baseline" not in out and out["config"]["parallelism"] == "iteration-sharded x2"
    # the whole build_analyze_model sequence on the shards (cleaning, T, solve, block validation): in the N > 1 line too
    assert out["full_pipeline"]["states"] > 2 and out["full_pipeline"]["total_seconds"] > 0 and len(out["full_pipeline"]["validation_JtargetSS"]) == 2
    assert "rank 0 trains on every rank's iterations" in out["config"]["training"]
    assert "torch.distributed" in r.stderr  # the stand-in was used (the oracle cannot create a communicator) ...
    assert out["config"]["exchange"].startswith("torch-fallback")   # ... and the line says so
    # without the explicit permission a run whose RCCL communicator cannot be created stops instead of measuring a stand-in
    env.pop("HAMSM_BENCH_ALLOW_FALLBACK")
    cmd[cmd.index("--master-port") + 1] = str(_free_port())
    r2 = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r2.returncode != 0 and "refusing to fall back" in r2.stderr
    # weak scaling on request: every rank its own full shard, rank 0 trains on its block; and rank-0 training of all shards
    env["HAMSM_BENCH_ALLOW_FALLBACK"] = "1"
    cmd[cmd.index("--master-port") + 1] = str(_free_port())
    r3 = subprocess.run(cmd + ["--scaling", "weak"], capture_output=True, text=True, env=env, timeout=600)
    assert r3.returncode == 0, r3.stderr[-2000:]
    o3 = json.loads([ln for ln in r3.stdout.splitlines() if ln.startswith("{")][0])
    assert o3["scaling"] == "weak" and o3["config"]["frames_total"] == 2 * out["config"]["frames_total"]
    assert "its own block" in o3["config"]["training"]
    cmd[cmd.index("--master-port") + 1] = str(_free_port())
    r4 = subprocess.run(cmd + ["--train-on", "strata"], capture_output=True, text=True, env=env, timeout=600)
    assert r4.returncode == 0, r4.stderr[-2000:]
    o4 = json.loads([ln for ln in r4.stdout.splitlines() if ln.startswith("{")][0])
    assert o4["scaling"] == "strong" and "stratum s trains on rank s % N" in o4["config"]["training"]
    assert abs(o4["value"]) > 0 and o4["config"]["frames_total"] == out["config"]["frames_total"]


def _worker4(rank, world, port, out, cuts):
    """world size 4, S = 4 strata dealt one per rank: an uneven split of the iterations, and one rank without any"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from msm_we_amd import _abi, synthetic
    from msm_we_amd.data import InMemorySource
    from msm_we_amd.model import modelWE

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synthetic.WEConfig("g4", 21, 26, 600, 8, 4, 6)
    lat = synthetic.latent_centres(cfg)
    h = _abi.Hamsm(lib=ctypes.CDLL(_build_oracle()))
    _gloo_comm(h, dist, rank, world)
    first, last = cuts[rank], cuts[rank + 1] - 1          # global iterations this rank discretizes (none: first > last)
    # (the reference's maxIter bookkeeping discretizes all but the last two loaded iterations, _data.py:860-866: the idle rank
    # loads two iterations and discretizes none)
    shard = {i - first + 1: synthetic.make_iteration(cfg, i, lat) for i in range(first, last + 3)}
    m = modelWE(h=h)
    m.set_iteration_source(InMemorySource(shard))
    basis, target = synthetic.basis_target_bounds(cfg.S)
    m.initialize(["g"], {"coords": None, "nAtoms": cfg.D, "coord_ndim": 1}, "g", basis_pcoord_bounds=basis, target_pcoord_bounds=target,
                 tau=1.0, pcoord_ndim=1)
    m.get_iterations()
    m.dimReduce()
    m.train_shards = "strata"
    bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
    m.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=7)
    st = dict(m.strata_exchange_stats)
    m.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=7)   # (cached exchange)
    m.get_fluxMatrix(0, first_iter=0, last_iter=m.maxIter)
    res = dict(labels=np.concatenate(m.dtrajs) if len(m.dtrajs) else np.zeros(0, dtype=np.int64), flux=m.fluxMatrixRaw,
               remap=[m.clusters.we_remap[b] for b in range(cfg.S)], steps=[getattr(cm, "n_steps_", -1) for cm in m.clusters.cluster_models],
               stats=[st["rows_sent"], st["rows_received"], st["rows_local"], st["rows_all_ranks"]])
    for b, cm in enumerate(m.clusters.cluster_models):
        res[f"centers{b}"], res[f"counts{b}"] = cm.cluster_centers_, cm._counts
    np.savez(out.format(rank), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_one_stratum_each_uneven_split_and_an_idle_rank(tmp_path, oracle_lib):
    """with `train_shards = "strata"` a rank is sent the rows of ITS strata only (one grouped
    exchange instead of R rooted gathers of everything).  World size 4, four strata, iterations split 9 / 0 / 12 / 4 (rank 1
    holds none): the model, every label and the flux matrix are the single process's; every rank received about a quarter of
    the training rows, not all of them."""
    import torch.multiprocessing as mp

    from msm_we_amd import _abi, synthetic
    from msm_we_amd.data import InMemorySource
    from msm_we_amd.model import modelWE

    cuts = [1, 10, 10, 22, 26]
    out = str(tmp_path / "rank{}.npz")
    mp.spawn(_worker4, args=(4, _free_port(), out, cuts), nprocs=4, join=True)
    r = [np.load(out.format(k)) for k in range(4)]
    cfg = synthetic.WEConfig("g4", 21, 26, 600, 8, 4, 6)
    lat = synthetic.latent_centres(cfg)
    h = _abi.Hamsm(lib=oracle_lib)
    c = modelWE(h=h)
    c.set_iteration_source(InMemorySource({i: synthetic.make_iteration(cfg, i, lat) for i in range(1, 28)}))
    basis, target = synthetic.basis_target_bounds(cfg.S)
    c.initialize(["g"], {"coords": None, "nAtoms": cfg.D, "coord_ndim": 1}, "g", basis_pcoord_bounds=basis, target_pcoord_bounds=target,
                 tau=1.0, pcoord_ndim=1)
    c.get_iterations()
    c.dimReduce()
    bm = RectilinearBinMapper([synthetic.strata_boundaries(cfg.S)])
    c.cluster_coordinates(cfg.K_s, streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=7)
    assert c.maxIter == 26 and np.array_equal(np.concatenate([x["labels"] for x in r]), np.concatenate(c.dtrajs))
    for b, cm in enumerate(c.clusters.cluster_models):
        for x in r:
            assert np.array_equal(x[f"centers{b}"], cm.cluster_centers_) and np.array_equal(x[f"counts{b}"], cm._counts), (b,)
    assert all(list(x["steps"]) == [cm.n_steps_ for cm in c.clusters.cluster_models] for x in r)
    assert all(list(x["remap"]) == [c.clusters.we_remap[b] for b in range(cfg.S)] for x in r)
    c.get_fluxMatrix(0, first_iter=0, last_iter=c.maxIter)
    for x in r[1:]:
        assert np.array_equal(x["flux"], r[0]["flux"])
    np.testing.assert_allclose(r[0]["flux"], c.fluxMatrixRaw, rtol=1e-12, atol=1e-300)
    # traffic: every training row is sent exactly once, every rank receives roughly its stratum's quarter
    stats = np.array([x["stats"] for x in r])
    assert stats[1, 2] == 0 and stats[1, 0] == 0 and stats[1, 1] > 0            # the idle rank sends nothing and still trains its stratum
    assert stats[:, 0].sum() == stats[:, 1].sum() <= stats[0, 3]
    assert (stats[:, 1] < 0.4 * stats[0, 3]).all()


def _worker_fail(rank, world, port, out):
    """a rank-local failure of the training phase must surface on EVERY rank, not strand the others in a collective"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from msm_we_amd import _abi

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, meta = load_golden("c1p_d10")
    h = _abi.Hamsm(lib=ctypes.CDLL(_build_oracle()))
    _gloo_comm(h, dist, rank, world)
    m = _shard_model(h, g, meta, (1, 20)[rank], (20, 38)[rank])
    m.train_shards = "strata"
    if rank == 1:
        def boom(*a, **k):
            raise MemoryError("rank-local failure in the middle of training")
        m._train_planned = boom
    bm = RectilinearBinMapper([np.array(meta["boundaries"], dtype=np.float32)])
    try:
        m.cluster_coordinates(meta["K_s"], streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=meta["seed"])
        msg = "no error"
    except Exception as e:  # noqa: BLE001
        msg = f"{type(e).__name__}: {e}"
    open(out.format(rank), "w").write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_local_training_failure_is_raised_on_every_rank(tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path / "rank{}.txt")
    mp.spawn(_worker_fail, args=(2, _free_port(), out), nprocs=2, join=True)
    m0, m1 = open(out.format(0)).read(), open(out.format(1)).read()
    assert m1.startswith("MemoryError") and m0.startswith("RuntimeError") and "another rank" in m0


def _worker_straddle(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from msm_we_amd import _abi

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, meta = load_golden("sparse_d10")
    h = _abi.Hamsm(lib=ctypes.CDLL(_build_oracle()))
    _gloo_comm(h, dist, rank, world)
    # rank 0 discretizes iterations 1..9, rank 1 iterations 10..58: the training call over iterations 9..11 lies on both
    m = _shard_model(h, g, meta, (1, 10)[rank], (10, 59)[rank])
    m.train_shards = "strata"
    bm = RectilinearBinMapper([np.array(meta["boundaries"], dtype=np.float32)])
    m.cluster_coordinates(meta["K_s"], streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=meta["seed"])
    res = dict(labels=np.concatenate(m.dtrajs), across=m.strata_exchange_stats["calls_across_shards"],
               remap=[m.clusters.we_remap[b] for b in range(2)])
    for b, cm in enumerate(m.clusters.cluster_models):
        res[f"centers{b}"], res[f"counts{b}"] = cm.cluster_centers_, cm._counts
    np.savez(out.format(rank), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_strata_shards_with_a_training_call_across_the_shard_boundary(tmp_path, oracle_lib):
    """golden `sparse_d10`: strata so thinly populated that a training call pools several iterations (`_clustering.py:664-716`).
    Cut inside such a call (iterations 9..11, shards 1..9 | 10..58): the frames a stratum indexes -- positions among the CALL's
    filtered rows applied to its unfiltered frames, quirk of `_clustering.py:874 / :892-896` -- then come from both ranks, and the
    model must still be the single process's (which equals the reference's golden)."""
    import torch.multiprocessing as mp

    from msm_we_amd import _abi

    out = str(tmp_path / "rank{}.npz")
    mp.spawn(_worker_straddle, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = np.load(out.format(0)), np.load(out.format(1))
    assert int(r0["across"]) >= 1 and int(r1["across"]) == int(r0["across"])
    g, meta = load_golden("sparse_d10")
    bm = RectilinearBinMapper([np.array(meta["boundaries"], dtype=np.float32)])
    h = _abi.Hamsm(lib=oracle_lib)
    c = _shard_model(h, g, meta, 1, 59)
    c.cluster_coordinates(meta["K_s"], streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=meta["seed"])
    assert np.array_equal(np.concatenate(c.dtrajs), g["dtrajs_post_cluster"])          # (the single process is the reference's run)
    assert np.array_equal(np.concatenate([r0["labels"], r1["labels"]]), np.concatenate(c.dtrajs))
    for b, cm in enumerate(c.clusters.cluster_models):
        for x in (r0, r1):
            assert np.array_equal(x[f"centers{b}"], cm.cluster_centers_) and np.array_equal(x[f"counts{b}"], cm._counts), b
    assert list(r0["remap"]) == list(r1["remap"]) == [c.clusters.we_remap[b] for b in range(2)]


def _worker_weighted(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from msm_we_amd import _abi
    from msm_we_amd.data import InMemorySource
    from msm_we_amd.model import modelWE

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, meta = load_golden("c1p_d10_weighted")
    h = _abi.Hamsm(lib=ctypes.CDLL(_build_oracle()))
    _gloo_comm(h, dist, rank, world)
    its = golden_iterations(g)
    first, last = (1, 14)[rank], (14, 38)[rank]
    m = modelWE(h=h)
    m.set_iteration_source(InMemorySource({i - first + 1: its[i] for i in range(first, last + 2)}))
    m.initialize(["g"], {"coords": None, "nAtoms": meta["D"], "coord_ndim": 1}, "g", basis_pcoord_bounds=meta["basis"],
                 target_pcoord_bounds=meta["target"], tau=meta["tau"], pcoord_ndim=1, use_weights_in_clustering=True)
    m.get_iterations()
    m.dimReduce()
    m.train_shards = "strata"
    bm = RectilinearBinMapper([np.array(meta["boundaries"], dtype=np.float32)])
    m.cluster_coordinates(meta["K_s"], streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=meta["seed"])
    np.savez(out.format(rank), labels=np.concatenate(m.dtrajs), **{f"centers{b}": cm.cluster_centers_ for b, cm in enumerate(m.clusters.cluster_models)},
             **{f"counts{b}": cm._counts for b, cm in enumerate(m.clusters.cluster_models)})
    dist.barrier()
    dist.destroy_process_group()


def test_strata_shards_with_sample_weights(tmp_path):
    """the weights travel with the frames they belong to (the frame a row's position selects, `_clustering.py:892-911`): the
    reference's weighted golden inputs, split 13 | 24 iterations over two ranks, give the single process's model and labels"""
    import torch.multiprocessing as mp

    out = str(tmp_path / "rank{}.npz")
    mp.spawn(_worker_weighted, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = np.load(out.format(0)), np.load(out.format(1))
    # one process on the same 37 discretized iterations (loaded 1..39), weights on
    from msm_we_amd import _abi
    from msm_we_amd.data import InMemorySource
    from msm_we_amd.model import modelWE

    g, meta = load_golden("c1p_d10_weighted")
    its = golden_iterations(g)
    h = _abi.Hamsm(lib=ctypes.CDLL(_build_oracle()))
    c = modelWE(h=h)
    c.set_iteration_source(InMemorySource({i: its[i] for i in range(1, 40)}))
    c.initialize(["g"], {"coords": None, "nAtoms": meta["D"], "coord_ndim": 1}, "g", basis_pcoord_bounds=meta["basis"],
                 target_pcoord_bounds=meta["target"], tau=meta["tau"], pcoord_ndim=1, use_weights_in_clustering=True)
    c.get_iterations()
    c.dimReduce()
    bm = RectilinearBinMapper([np.array(meta["boundaries"], dtype=np.float32)])
    c.cluster_coordinates(meta["K_s"], streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=meta["seed"])
    assert np.array_equal(np.concatenate([r0["labels"], r1["labels"]]), np.concatenate(c.dtrajs))
    for b, cm in enumerate(c.clusters.cluster_models):
        for x in (r0, r1):
            assert np.array_equal(x[f"centers{b}"], cm.cluster_centers_) and np.array_equal(x[f"counts{b}"], cm._counts), b


def _worker_case(rank, world, port, out, case, cut):
    """golden `case` split at global iteration `cut` over two ranks, `train_shards = "strata"`, the case's own mapper / start"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from msm_we_amd import _abi
    from msm_we_amd.data import InMemorySource
    from msm_we_amd.model import modelWE

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, meta = load_golden(case)
    h = _abi.Hamsm(lib=ctypes.CDLL(_build_oracle()))
    _gloo_comm(h, dist, rank, world)
    its = golden_iterations(g)
    n_all = max(its) - 2                                   # the single process discretizes 1..n_all (loaded n_all + 2)
    first, last = (1, cut)[rank], (cut - 1, n_all)[rank]
    m = modelWE(h=h)
    m.set_iteration_source(InMemorySource({i - first + 1: its[i] for i in range(first, last + 3)}))
    m.initialize(["g"], {"coords": None, "nAtoms": meta["D"], "coord_ndim": 1}, "g", basis_pcoord_bounds=meta["basis"],
                 target_pcoord_bounds=meta["target"], tau=meta["tau"], pcoord_ndim=meta.get("pcoord_ndim", 1),
                 use_weights_in_clustering=meta["use_weights"])
    m.get_iterations()
    m.dimReduce()
    m.train_shards = "strata"
    if meta.get("voronoi"):
        from msm_we_amd.binning import VoronoiBinMapper, euclidean_dfunc

        bm = VoronoiBinMapper(euclidean_dfunc, np.array(meta["voronoi"], dtype=np.float32))
    else:
        bm = RectilinearBinMapper([np.array(meta["boundaries"], dtype=np.float32)])
    kw = {"first_cluster_iter": meta["first_iter"]} if meta.get("first_iter", 1) != 1 else {}
    m.cluster_coordinates(meta["K_s"], streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm, random_state=meta["seed"], **kw)
    np.savez(out.format(rank), labels=np.concatenate(m.dtrajs), remap=[m.clusters.we_remap[b] for b in range(bm.nbins)])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,cut", [("late_start_d10", 12), ("voronoi_d10", 9), ("gap_d10", 15)])
def test_strata_shards_reproduce_the_reference_goldens(tmp_path, case, cut):
    """two ranks, stratum-sharded training, against the REFERENCE's own discretization of the same run: a late first training
    iteration (`first_cluster_iter`, global numbering), a Voronoi mapper with a never-visited bin, four strata with a gap"""
    import torch.multiprocessing as mp

    out = str(tmp_path / "rank{}.npz")
    mp.spawn(_worker_case, args=(2, _free_port(), out, case, cut), nprocs=2, join=True)
    r0, r1 = np.load(out.format(0)), np.load(out.format(1))
    g, meta = load_golden(case)
    assert np.array_equal(np.concatenate([r0["labels"], r1["labels"]]), g["dtrajs_post_cluster"])
    assert list(r0["remap"]) == list(r1["remap"])

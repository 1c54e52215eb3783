"""The in-library RCCL path with TWO ranks (it had only ever run on a 1-rank communicator and over gloo stand-ins).
Needs two GPUs on one node: skipped on the single-GPU boxes the suite normally runs on; there `test_gpu_parity.py::
test_rccl_collectives_on_one_rank_communicator` covers the same entry points on one rank, and `test_distributed_gloo.py` the
N > 1 logic.  Every check is one the ranks can only pass TOGETHER: all-reduces, the model broadcast per stratum from different
roots, the rooted gather, the grouped exchange -- and the collective argument checks (different roots, size tables that do not
match): both ranks must come back with the same error instead of one waiting inside the collective."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _n_gpus():
    try:
        import torch

        return torch.cuda.device_count()      # (does not initialise a device)
    except Exception:  # noqa: BLE001
        return 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from msm_we_amd import _abi

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)        # bootstrap only: the 128-byte RCCL id
    h = _abi.Hamsm(device=rank)
    uid = [h.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    h.comm_init(world, rank, uid[0])
    log = []
    rng = np.random.default_rng(7)
    # all-reduces
    v = rng.normal(size=4097) + rank
    d = h.to_device(v)
    h.allreduce_sum_f64(d, v.size)
    base = np.random.default_rng(7).normal(size=4097)
    log.append(("sum", bool(np.array_equal(d.to_host(), (base + 0) + (base + 1)))))
    # a model whose strata live on different ranks
    D, S, K = 6, 3, 5
    h.set_geometry(D, 1, S)
    h.set_mapper_rectilinear([np.array([0, 1, 2, np.inf], dtype=np.float32)])
    h.set_bounds(np.array([[-np.inf, -1.0]]), np.array([[9.0, np.inf]]))
    cents = {s: np.random.default_rng(100 + s).normal(size=(K, D)) for s in range(S)}
    roots = np.array([0, 1, 0], dtype=np.int32)
    for s in range(S):
        if roots[s] == rank:
            h.set_centroids(s, cents[s])
    h.set_remap(np.arange(S, dtype=np.int32))
    h.bcast_strata(roots)
    log.append(("strata", all(np.array_equal(h.get_centroids(s)[0], cents[s]) for s in range(S))))
    # ranks that name different roots: the same error on both, nothing hangs, the communicator stays usable
    try:
        h.bcast_strata(np.array([0, rank, 0], dtype=np.int32))
        log.append(("roots", False))
    except _abi.HamsmError as e:
        log.append(("roots", "DIFFERENT roots" in str(e) or "could not prepare" in str(e)))
    h.bcast_strata(roots)
    # rooted gather and grouped exchange
    mine = (np.arange(1000 + 10 * rank, dtype=np.float64) + 1e6 * rank)
    d_s = h.to_device(mine)
    sizes = [8 * 1000, 8 * 1010]
    d_r = h.zeros(2010, np.float64) if rank == 0 else None
    h.gather_to_root(d_s, mine.nbytes, d_r, sizes, 0)
    h.sync()
    if rank == 0:
        log.append(("gather", bool(np.array_equal(d_r.to_host(), np.concatenate([np.arange(1000.0), np.arange(1010.0) + 1e6])))))
    send = [8 * 300, 8 * (len(mine) - 300)] if rank == 0 else [8 * 500, 8 * (len(mine) - 500)]
    recv = [8 * 300, 8 * 500] if rank == 0 else [8 * 700, 8 * 510]
    d_x = h.zeros(sum(recv) // 8, np.float64)
    h.exchange(d_s, send, d_x, recv)
    h.sync()
    other = np.arange(1000 + 10 * (1 - rank), dtype=np.float64) + 1e6 * (1 - rank)
    want = (np.concatenate([mine[:300], other[:500]]) if rank == 0 else np.concatenate([other[300:], mine[500:]]))
    log.append(("exchange", bool(np.array_equal(d_x.to_host(), want))))
    try:     # a size table the other rank does not expect: every rank returns the same error
        bad = list(send)
        bad[1 - rank] -= 8 * rank
        h.exchange(d_s, bad, d_x, recv)
        log.append(("tables", rank == 0 and False))
    except _abi.HamsmError as e:
        log.append(("tables", "expects" in str(e)))
    h.allreduce_sum_f64(d, 1)     # still usable
    h.sync()
    open(out.format(rank), "w").write(repr(log))
    dist.barrier()
    h.comm_destroy()
    h.close()
    dist.destroy_process_group()


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs on one node (the in-library RCCL path with N > 1)")
def test_two_rank_rccl_collectives(tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path / "rank{}.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in range(2):
        log = eval(open(out.format(r)).read())     # noqa: S307 -- our own repr
        assert all(ok for _, ok in log), (r, log)

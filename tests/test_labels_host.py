"""`dtrajs` / `pair_dtrajs` (the reference's host-side post-condition, `_clustering.py:1206-1240`) are views of page-locked
process-level blocks (`_abi.HostBlock`, ABI 9): materialised lazily and one by one, equal to the golden discretization, alive past
the context that produced them, and the blocks are recycled once no view is left."""
import gc

import numpy as np
import pytest

from conftest import build_model, load_golden
from msm_we_amd import _abi


@pytest.mark.parametrize("backend", ["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def test_dtrajs_are_lazy_views_that_outlive_the_context(request, backend):
    lib = request.getfixturevalue("oracle_lib" if backend == "oracle" else "hip_lib")
    g, meta = load_golden("c1p_d10")
    gc.collect()
    _abi.HostBlock.drain()            # (earlier tests of the session may have left blocks of this very size in the pool)
    h = _abi.Hamsm(lib=lib)
    m, _ = build_model(h, g, meta)
    assert m._dtrajs_cache is None and m._pairs_cache is None       # nothing crossed the link yet
    dtr = m.dtrajs
    assert m._pairs_cache is None                                    # `dtrajs` alone does not fetch the pairs
    assert all(d.dtype == np.int64 for d in dtr) and np.array_equal(np.concatenate(dtr), g["dtrajs_post_cluster"])
    pairs = m.pair_dtrajs
    assert np.array_equal(np.concatenate(pairs), g["pairs_post_cluster"]) and pairs[0].shape[1] == 2
    assert m.dtrajs is dtr                                            # cached: no second copy
    base = dtr[0]
    while isinstance(base.base, np.ndarray):
        base = base.base
    assert isinstance(base.base, _abi.HostBlock)
    addr = base.base.ptr
    keep = [d.copy() for d in dtr]
    del m
    h.close()                                                        # the views stay valid: the block belongs to the process
    gc.collect()
    assert all(np.array_equal(a, b) for a, b in zip(dtr, keep))
    # once the last view is gone the block returns to the pool and serves the next request of its size
    nbytes = base.base.nbytes
    del dtr, base, pairs
    gc.collect()
    assert addr in _abi.HostBlock._pool.get((lib._handle, nbytes), [])
    again = _abi.HostBlock.array(lib, nbytes // 8, np.int64)
    assert again.ctypes.data == addr
    del again
    gc.collect()
    _abi.HostBlock.drain()
    assert not any(_abi.HostBlock._pool.values()) and _abi.HostBlock._pooled_bytes == 0

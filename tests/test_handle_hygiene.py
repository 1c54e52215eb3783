"""Handle and buffer lifetime rules of the C ABI (round-2 fix of the zero-filled-context failure).

A context handle stays recognisable after hamsm_destroy (the struct is never returned to the heap), device blocks belong
to the context that allocated them, and every entry point validates its handle before touching it.  CPU part: the oracle
(also run under AddressSanitizer by oracle/run_asan.sh).  GPU part: the same rules on the HIP library."""
import ctypes
import gc

import numpy as np
import pytest

from msm_we_amd import _abi


def _rules(lib):
    lib = _abi.bind(lib)
    h = _abi.Hamsm(lib=lib)
    raw = ctypes.c_void_p(h.ctx.value)
    h.set_geometry(4, 1, 2)
    buf = h.alloc(16, np.float64)
    ptr = buf.ptr
    # a block is freed once; the second free and a foreign pointer are errors, not heap operations
    buf.free()
    assert lib.hamsm_free(raw, ptr) == -1
    assert b"hamsm_malloc" in lib.hamsm_last_error(raw)
    other = _abi.Hamsm(lib=lib)
    b2 = other.alloc(8, np.float64)
    assert lib.hamsm_free(raw, b2.ptr) == -1          # a block of ANOTHER context
    keep = h.alloc(32, np.float64)                    # still live when its context goes away
    h.close()
    # stale handle: every entry point refuses it, with a message that says so; destroy twice is a no-op
    assert lib.hamsm_set_geometry(raw, 4, 1, 2) == -1
    assert b"stale handle" in lib.hamsm_last_error(raw)
    assert lib.hamsm_sync(raw) == -1
    lib.hamsm_destroy(raw)
    lib.hamsm_destroy(raw)
    # contexts created afterwards never alias the dead one
    fresh = [_abi.Hamsm(lib=lib) for _ in range(8)]
    assert all(f.ctx.value != raw.value for f in fresh)
    assert lib.hamsm_set_geometry(raw, 4, 1, 2) == -1
    for f in fresh:
        f.set_geometry(3, 1, 1)
        f.close()
    # a wrapper that outlives its context frees nothing
    keep.free()
    del keep, buf
    gc.collect()
    # a pointer that never was a context
    junk = ctypes.create_string_buffer(4096)
    assert lib.hamsm_sync(ctypes.cast(junk, ctypes.c_void_p)) == -1
    assert b"not a hamsm context" in lib.hamsm_last_error(ctypes.cast(junk, ctypes.c_void_p))
    other.set_geometry(4, 1, 2)                        # the bystander context is untouched
    # a context whose memory somebody else wrote over says so (the round-1 failure would now read like this)
    magic = ctypes.c_uint64.from_address(other.ctx.value)
    saved, magic.value = magic.value, 0
    assert lib.hamsm_sync(other.ctx) == -1 and b"OVERWRITTEN" in lib.hamsm_last_error(other.ctx)
    magic.value = saved
    other.sync()
    other.close()


def test_handle_rules_oracle(oracle_lib):
    _rules(oracle_lib)


@pytest.mark.gpu
def test_handle_rules_hip(hip_lib):
    _rules(hip_lib)


def _setup(h, D, S, K_s, rng):
    h.set_geometry(D, 1, S)
    b = np.arange(S + 1, dtype=np.float32)
    b[-1] = np.inf
    h.set_mapper_rectilinear([b])
    h.set_bounds([[-np.inf, 0.05]], [[S - 0.05, np.inf]])
    cents = []
    for s in range(S):
        Cm = rng.normal(size=(K_s[s], D)) * 3
        h.set_centroids(s, Cm, csq=np.einsum("ij,ij->i", Cm, Cm))
        cents.append(Cm)
    return cents


@pytest.mark.parametrize("seed", range(40))
def test_assign_fuzz_oracle_contexts_interleaved(oracle_lib, seed):
    """the shape fuzz of the GPU suite (test_assign_random_shapes_hip_equals_oracle) on two oracle contexts that are created,
    configured and destroyed interleaved, with wrappers dropped at arbitrary points: labels agree with a NumPy restatement and
    no context ever loses its state.  This is the test oracle/run_asan.sh runs under AddressSanitizer."""
    rng = np.random.default_rng(7000 + seed)
    D, S = int(rng.integers(1, 129)), int(rng.integers(1, 6))
    K_s = [int(rng.integers(1, 300)) for _ in range(S)]
    dtype = np.float32 if rng.integers(2) else np.float64
    n = int(rng.integers(1, 1500))
    a = _abi.Hamsm(lib=oracle_lib)
    r = np.random.default_rng(seed)
    cents = _setup(a, D, S, K_s, r)
    b = _abi.Hamsm(lib=oracle_lib)          # a second context is born while the first is configured
    _setup(b, max(1, D // 2), S, [max(1, k // 2) for k in K_s], np.random.default_rng(seed + 1))
    tmp = [a.alloc(int(rng.integers(1, 2000)), np.float64) for _ in range(4)]
    X = (r.normal(size=(n, D)) * 3).astype(dtype)
    p = r.uniform(0, S, size=(n, 1))
    b.close()                                # ... and dies before the first is used
    del tmp[1]
    gc.collect()
    lab = a.predict_host(X, p)
    Kfit = sum(K_s)
    off = np.concatenate([[0], np.cumsum(K_s)])
    for i in range(0, n, max(1, n // 50)):
        pi = p[i, 0]
        if pi > S - 0.05:
            assert lab[i] == Kfit + 1
        elif pi < 0.05:
            assert lab[i] == Kfit
        else:
            s = min(int(np.float32(pi)), S - 1)
            Cm = cents[s]
            d = np.einsum("ij,ij->i", Cm, Cm) - 2 * Cm @ X[i].astype(np.float64)
            j = lab[i] - off[s]
            assert 0 <= j < K_s[s] and d[j] <= d.min() + 1e-9 * max(1.0, abs(d.min()))
    a.close()


def test_seed_chosen_checks_its_buffer_and_the_arena(oracle_h):
    """`hamsm_kmeans_seed_chosen` copies the item's recorded K indices: a smaller caller buffer is an error, not an
    overflow (the HIP library additionally refuses once another call has taken the scratch arena the indices live in:
    tests/test_gpu_parity.py)"""
    import numpy as np

    from msm_we_amd import _abi

    h = oracle_h
    h.set_geometry(4, 1, 1)
    rng = np.random.default_rng(0)
    X = rng.normal(size=(200, 4)).astype(np.float32)
    d_X = h.to_device(X)
    K, L = 6, 3
    h.kmeans_seed_multi([0], d_X, _abi.F32, 4, [0], None, [0], [200], None, [None], [K], [5], [rng.uniform(size=(K - 1) * L)], [L])
    assert len(h.kmeans_seed_chosen(0, K)) == K
    with pytest.raises(_abi.HamsmError, match="holds"):
        h.kmeans_seed_chosen(0, K - 1)

"""hamsm_assign_pair (ABI 10): both sides of the same segments in one call -- the reference's discretization task predicts the
parent frames and then the child frames of an iteration (`_clustering.py:1299-1301`, `:1315-1316`).  Same labels, same `seen`
marks and the same errors as two hamsm_assign calls; on the oracle under `-m "not gpu"`, on the HIP library under `-m gpu`."""
import numpy as np
import pytest

from msm_we_amd import _abi
from msm_we_amd._abi import HamsmError

from test_gpu_parity import _frames, _setup


@pytest.mark.parametrize("D,S,K_s,dtype,n", [(30, 4, 250, np.float32, 20011), (13, 3, 25, np.float64, 5003), (64, 4, 77, np.float32, 1)])
def test_pair_equals_two_calls(any_h, D, S, K_s, dtype, n):
    h = any_h
    rng = np.random.default_rng(7)
    _setup(h, D, S, K_s, rng)
    Xp, p0 = _frames(rng, n, D, S, dtype)
    Xc, p1 = _frames(rng, n, D, S, dtype)
    dt = _abi.F32 if dtype == np.float32 else _abi.F64
    d = [h.to_device(a) for a in (Xp, Xc, p0, p1)]
    out = []
    for pair in (False, True):
        la, lb, seen = h.alloc(n, np.int32), h.alloc(n, np.int32), h.zeros((2, S), np.int32)
        if pair:
            h.assign_pair(d[0], d[1], dt, D, d[2], d[3], n, la, lb, seen)
        else:
            h.assign(d[0], dt, D, d[2], n, la, seen)
            h.assign(d[1], dt, D, d[3], n, lb, seen)
        out.append((la.to_host(), lb.to_host(), seen.to_host()))
        for b in (la, lb, seen):
            b.free()
    for b in d:
        b.free()
    assert all(np.array_equal(x, y) for x, y in zip(*out))
    assert not np.array_equal(out[0][0], out[0][1])  # (the sides differ: each got its own frames)


def test_pair_reports_the_parent_sides_error_first_and_refuses_an_external_mapper(any_h):
    h = any_h
    rng = np.random.default_rng(2)
    _setup(h, 10, 2, 8, rng, fit=[True, False])
    n = 300
    X, p = _frames(rng, n, 10, 2, np.float32)
    p_ok = np.full((n, 1), 0.5)                 # stratum 0 only: fitted
    p_out = p_ok.copy()
    p_out[17, 0] = -3.0                         # outside the bin space -> HAMSM_ERANGE
    d_X, d_p, d_ok, d_out = (h.to_device(a) for a in (X, p, p_ok, p_out))
    la, lb = h.alloc(n, np.int32), h.alloc(n, np.int32)
    h.assign_pair(d_X, d_X, _abi.F32, 10, d_ok, d_ok, n, la, lb)            # no `seen`: allowed, as with hamsm_assign
    assert np.array_equal(la.to_host(), lb.to_host())
    with pytest.raises(HamsmError) as e:                                   # parent side unfitted (-4), child side out of range (-5)
        h.assign_pair(d_X, d_X, _abi.F32, 10, d_p, d_out, n, la, lb)
    assert e.value.code == -4
    with pytest.raises(HamsmError) as e:                                   # parent side fine, child side out of range
        h.assign_pair(d_X, d_X, _abi.F32, 10, d_ok, d_out, n, la, lb)
    assert e.value.code == -5
    h.assign_pair(d_X, d_X, _abi.F32, 10, d_ok, d_ok, n, la, lb)            # the error does not stick
    h.assign_pair(d_X, d_X, _abi.F32, 10, d_ok, d_ok, 0, la, lb)            # nothing to do
    # an external mapper's frame bins are set per side
    h.set_mapper_external()
    h.set_frame_bins(h.to_device(np.zeros(n, dtype=np.int32)))
    with pytest.raises(HamsmError) as e:
        h.assign_pair(d_X, d_X, _abi.F32, 10, d_ok, d_ok, n, la, lb)
    assert e.value.code == -1

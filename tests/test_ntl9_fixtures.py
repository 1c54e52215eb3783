"""The reference's own NTL9 fixtures (tests/golden/ntl9, copied data files; reference tests
tests/test_msm_we.py:148-180): fluxmatrix.npy -> tmatrix.npy -> pSS.npy -> JtargetSS.npy, and the
strongly-connected-set size implied by fluxmatrix_raw.npy -> 192 kept clusters.
Every test that touches the library runs on the oracle and, under `-m gpu`, on the HIP library (`any_h`): `hamsm_tmatrix_csr` on the
GPU against the reference's own `tmatrix.npy`, bit for bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

NTL9 = os.path.join(GOLDEN, "ntl9")


def _model(h):
    from msm_we_amd.model import modelWE

    m = modelWE(h=h)
    m.fluxMatrix = np.load(os.path.join(NTL9, "fluxmatrix.npy"))
    m.indBasis, m.indTargets, m.nBins, m.tau = np.array([192]), np.array([193]), 194, 1e-9
    return m


def test_get_tmatrix(any_h):
    m = _model(any_h)
    m.get_Tmatrix()
    ref = np.load(os.path.join(NTL9, "tmatrix.npy"))
    assert np.array_equal(m.Tmatrix, ref)  # reference test uses np.allclose; we are bit-identical
    assert m.Tmatrix_csr.nnz == np.count_nonzero(ref)


@pytest.mark.parametrize("dense_eig_max, dense_lu_density", [(10 ** 9, 0.10), (0, 0.10), (0, 2.0), (10 ** 9, 2.0)],
                         ids=["eig+lapack", "arpack+lapack", "arpack+superlu", "eig+superlu"])
def test_get_steady_state_and_target_flux(any_h, monkeypatch, dense_eig_max, dense_lu_density):
    """every solver combination the host solve can pick (dense eig = the reference's chain; ARPACK seed above
    `dense_eig_max` states; LAPACK or SuperLU inverse iteration by matrix density) lands on the reference's numbers"""
    from msm_we_amd import analysis

    monkeypatch.setattr(analysis, "DENSE_LU_DENSITY", dense_lu_density)
    m = _model(any_h)
    m.dense_eig_max = dense_eig_max
    m.get_Tmatrix()
    m.get_steady_state()
    ref = np.load(os.path.join(NTL9, "pSS.npy"))
    np.testing.assert_allclose(m.pSS, ref, rtol=1e-6, atol=1e-300)
    m.get_steady_state_target_flux()
    J = float(np.load(os.path.join(NTL9, "JtargetSS.npy")))
    assert abs(m.JtargetSS - J) <= 1e-6 * J


def test_connected_set_of_raw_fixture():
    from msm_we_amd.clustering import find_connected_sets

    F = np.load(os.path.join(NTL9, "fluxmatrix_raw.npy")).copy()
    assert F.shape == (302, 302)
    F[-1, -2] = 1.0
    cc = find_connected_sets(F, directed=True)
    removed = sum(len(c) for c in cc[1:])
    assert 300 - removed == 192  # organized.obj: n_clusters == 192 (quirk Q9)


def test_committor_bounds(any_h):
    m = _model(any_h)
    m.get_committor(conv=1e-5)
    assert m.q[192] == 0.0 and m.q[193] == 1.0 and np.all((m.q >= 0) & (m.q <= 1 + 1e-12))

"""The C-ABI libraries load on a GPU-less host and export every symbol include/hamsm.h declares
(no compute calls).  The product library must refuse to create a context without a GPU -- there is
no CPU path behind it."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "hamsm.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hamsm_[a-z0-9_]+)\s*\(", src)))


def test_binding_table_covers_header():
    from msm_we_amd import _abi

    assert sorted(_abi.SYMBOLS) == _declared()


def test_abi_version_is_one_number(oracle_lib):
    """header, binding table, both libraries and the driver's build() check agree on the ABI version"""
    from msm_we_amd import _abi

    v = int(re.search(r"#define\s+HAMSM_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert _abi.ABI_VERSION == v and oracle_lib.hamsm_abi_version() == v
    assert _abi.bind(_abi.load()).hamsm_abi_version() == v
    assert "_abi.ABI_VERSION" in open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert np.dtype(_abi.Hamsm.STEP_DTYPE).itemsize == 48  # sizeof(hamsm_step)


def test_oracle_exports_abi(oracle_lib):
    for name in _declared():
        assert hasattr(oracle_lib, name), name
    oracle_lib.hamsm_backend_name.restype = ctypes.c_char_p
    assert oracle_lib.hamsm_backend_name() == b"cpu-oracle"


def test_hip_library_exports_abi_and_has_no_cpu_path():
    from msm_we_amd import _abi

    so = _abi.library_path()
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.dirname(so)])
    lib = _abi.bind(_abi.load())
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.hamsm_backend_name() == b"hip-gfx950"
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        has_gpu = os.path.exists("/dev/kfd")
    if not has_gpu:
        with pytest.raises(_abi.HamsmError):
            _abi.Hamsm(lib=lib)  # hamsm_create fails loudly: HAMSM_EDEVICE


def test_loader_fails_loudly_when_library_is_missing(monkeypatch, tmp_path):
    from msm_we_amd import _abi

    monkeypatch.setattr(_abi, "_HERE", str(tmp_path))
    with pytest.raises(ImportError):
        _abi.load()

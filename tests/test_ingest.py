"""a15 / f2: the ingest side of the path -- the WESTPA HDF5 reader (against a stand-in `h5py`: the real one is not in the
image), the per-iteration .npz fallback, and the streaming double-buffered staging (several blocks, float32 fallback to
float64, NaN bookkeeping) -- all must give the model the in-memory source gives."""
import sys
import types

import numpy as np
import pytest

from conftest import build_model, golden_iterations, load_golden
from msm_we_amd.data import InMemorySource, NpzDirectorySource


class _FakeNode(dict):
    """h5py.Group / Dataset stand-in: nested keys with '/', datasets sliced with [:] and field names"""

    def __getitem__(self, key):
        if isinstance(key, str) and "/" in key:
            head, rest = key.split("/", 1)
            return dict.__getitem__(self, head)[rest]
        return dict.__getitem__(self, key)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False


class _FakeDataset:
    def __init__(self, a):
        self.a = a
        self.reads = 0

    def __getitem__(self, k):
        self.reads += 1
        return self.a[k]

    def __array__(self, dtype=None, copy=None):
        self.reads += 1
        return np.asarray(self.a, dtype=dtype)


def _fake_h5py(files):
    opened = []

    class File(_FakeNode):
        def __init__(self, path, mode="r"):
            super().__init__(files[path])
            opened.append(path)

        def close(self):
            pass

    return types.SimpleNamespace(File=File, opened=opened)


def _west_h5_tree(iters, n_atoms, split=None):
    """the layout of augmentation_driver.py:173-180 for a dict of iterations (features reshaped to (nAtoms, 3) coordinates)"""
    tree = _FakeNode(iterations=_FakeNode())
    for it, d in iters.items():
        n = len(d["w"])
        sel = slice(None) if split is None else split(n)
        coord = np.stack([d["Xp"][sel].reshape(-1, n_atoms, 3), d["Xc"][sel].reshape(-1, n_atoms, 3)], axis=1).astype(np.float32)
        pcoord = np.stack([d["p0"][sel], d["p1"][sel]], axis=1)  # (nSeg, pcoord_len = 2, ndim)
        seg_index = np.zeros(coord.shape[0], dtype=[("weight", "<f8"), ("parent_id", "<i8")])
        seg_index["weight"] = d["w"][sel]
        tree["iterations"]["iter_%08d" % it] = _FakeNode(pcoord=_FakeDataset(pcoord), seg_index=_FakeDataset(seg_index),
                                                           auxdata=_FakeNode(coord=_FakeDataset(coord)))
    return tree


def test_westpa_h5_source_reads_the_reference_layout(oracle_h, monkeypatch):
    from msm_we_amd import data

    g, meta = load_golden("c1p_d30")            # D = 30 = 10 atoms x 3
    iters = golden_iterations(g)
    fake = _fake_h5py({"west.h5": _west_h5_tree(iters, 10)})
    monkeypatch.setitem(sys.modules, "h5py", fake)
    src = data.WestpaH5Source("west.h5", auxpath="coord", pcoord_ndim=1)
    assert src.iterations() == sorted(iters) and fake.opened == ["west.h5"]     # opened once, not per iteration
    d7 = src.load(7)
    assert d7["parent"].shape == (len(iters[7]["w"]), 10, 3) and d7["parent"].dtype == np.float32
    assert np.array_equal(d7["child"].reshape(-1, 30), iters[7]["Xc"]) and np.array_equal(d7["w"], iters[7]["w"])
    m, st = build_model(oracle_h, g, meta, source=src)     # flatten featurizer (the notebook's) over (nSeg, 10, 3)
    assert np.array_equal(np.concatenate(m.dtrajs), g["dtrajs_post_cluster"])
    assert m._store.x_dtype == 0 and m.ingest_stats["mode"] == "host"
    # two files, every iteration's segments split between them (the reference's fileList: segments are concatenated)
    fake2 = _fake_h5py({"a.h5": _west_h5_tree(iters, 10, lambda n: slice(0, n // 2)),
                        "b.h5": _west_h5_tree(iters, 10, lambda n: slice(n // 2, None))})
    monkeypatch.setitem(sys.modules, "h5py", fake2)
    src2 = data.WestpaH5Source(["a.h5", "b.h5"])
    assert np.array_equal(src2.load(3)["child"].reshape(-1, 30), iters[3]["Xc"])
    src.close()
    src2.close()


def test_h5_source_without_h5py_says_so(monkeypatch):
    from msm_we_amd import data

    monkeypatch.setitem(sys.modules, "h5py", None)
    with pytest.raises(ImportError, match="h5py"):
        data.WestpaH5Source("west.h5")


@pytest.mark.parametrize("case", ["c1p_d10", "nan_d10"])
def test_streaming_staging_in_many_blocks(oracle_h, tmp_path, monkeypatch, case):
    """blocks of ~3 iterations through the two pinned slabs: the staged arrays are those of a single-block staging"""
    from msm_we_amd.model import modelWE

    g, meta = load_golden(case)
    src = NpzDirectorySource.export(InMemorySource(golden_iterations(g)), tmp_path / "iters")
    stores = []
    for chunk_bytes in (64 << 20, 3 * 128 * (2 * 10 * 4 + 24)):
        monkeypatch.setattr(modelWE, "STAGE_CHUNK_BYTES", chunk_bytes)
        m, _ = build_model(oracle_h, g, meta, source=src)
        st = m._store
        stores.append([st.Xp.to_host(), st.Xc.to_host(), st.p0.to_host(), st.p1.to_host(), st.w.to_host(), st.w_flux.to_host(),
                       np.concatenate(m.dtrajs), m.ingest_stats["blocks"]])
    assert stores[0][-1] == 1 and stores[1][-1] > 10
    for a, b in zip(stores[0][:-1], stores[1][:-1]):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(stores[1][6], g["dtrajs_post_cluster"])


def test_staging_falls_back_to_float64_when_float32_would_lose_bits(oracle_h):
    g, meta = load_golden("c1p_d10")
    its = {k: {a: np.array(b, dtype=np.float64 if a in ("Xp", "Xc") else b.dtype) for a, b in v.items()}
           for k, v in golden_iterations(g).items()}
    its[23]["Xc"][5, 2] += 1e-12        # one value that float32 cannot hold, late in the run
    m, _ = build_model(oracle_h, g, meta, source=InMemorySource(its), seed_forcing=False)
    assert m._store.x_dtype == 1 and m._store.Xc.to_host().reshape(-1, 10)[m._store.offsets[22] + 5, 2] == its[23]["Xc"][5, 2]

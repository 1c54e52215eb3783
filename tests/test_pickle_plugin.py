"""Level B1-B3 of the boundary: the model and the cluster object survive pickle (reference:
`_clustering.py:195,1165,1171`, tests/fixtures/hamsms.py:13-56 pickles the model after every stage), and the WESTPA plugin
shim builds a model through a stub sim_manager (`westpa_plugins/hamsm_driver.py:35-144`)."""
import copy
import pickle

import numpy as np
import pytest

from conftest import RectilinearBinMapper, build_model, golden_iterations, load_golden
from msm_we_amd import _abi


def _roundtrip(h_first, h_second):
    g, meta = load_golden("c1p_d10")
    m, _ = build_model(h_first, g, meta, seed_forcing=False)
    m.get_fluxMatrix(0, first_iter=1, last_iter=m.maxIter)
    F0 = m.fluxMatrixRaw.copy()
    dtr0 = [d.copy() for d in m.dtrajs]
    # the cluster object pickles in the state every consumer sees it in (.model attached, ctx bound)
    blob_c = pickle.dumps(m.clusters)
    blob_m = pickle.dumps(m)
    assert b"CDLL" not in blob_m and len(blob_m) < 4_000_000
    with pytest.raises(TypeError):
        pickle.dumps(h_first)            # a live context refuses, with a message that says what to do
    assert copy.deepcopy(h_first) is h_first
    cl = pickle.loads(blob_c)
    for a, b in zip(cl.cluster_models, m.clusters.cluster_models):
        assert a.fitted == b.fitted
        if a.fitted:
            assert np.array_equal(a.cluster_centers_, b.cluster_centers_)
    assert dict(cl.we_remap) == dict(m.clusters.we_remap)
    with pytest.raises(RuntimeError):    # no silent CPU path: predict without a context fails loudly
        cl.predict(np.zeros((1, meta["D"])))
    m2 = pickle.loads(blob_m)
    assert m2._h is None and m2._store is None and m2.clusters.model is m2
    assert all(np.array_equal(a, b) for a, b in zip(m2.dtrajs, dtr0))      # discretization materialised in the pickle
    assert np.array_equal(m2.fluxMatrixRaw, F0)
    # bind a (different) context: predict, flux, cleaning and the solve run again on re-staged frames
    m2.attach(h_second)
    it = 7
    m2.load_iter_data(it)
    d = golden_iterations(g)[it]
    lab = m2.clusters.predict(d["Xc"])
    assert np.array_equal(lab, dtr0[it - 1])
    m2.get_fluxMatrix(0, first_iter=1, last_iter=m2.maxIter)
    assert np.array_equal(m2.fluxMatrixRaw, F0)
    m2.organize_fluxMatrix()
    assert np.array_equal(np.concatenate(m2.dtrajs), g["dtrajs_clean"])
    assert np.array_equal(m2.fluxMatrix, g["fluxMatrix"])
    # ... and the organised model pickles again (stage snapshots: hamsms.py "organized")
    m3 = pickle.loads(pickle.dumps(m2))
    assert np.array_equal(m3.fluxMatrix, g["fluxMatrix"]) and m3.nBins == m2.nBins
    # the standalone cluster object, attached to a context and a model, predicts too (ray.put(clusters) consumer)
    cl.attach(h_second)
    cl.model = m2
    # (cl holds the pre-clean centroids: compare with the pre-clean discretization)
    m2.load_iter_data(it)
    assert np.array_equal(cl.predict(d["Xc"]), dtr0[it - 1])


def test_model_and_clusters_pickle_roundtrip_oracle(oracle_lib):
    a, b = _abi.Hamsm(lib=oracle_lib), _abi.Hamsm(lib=oracle_lib)
    try:
        _roundtrip(a, b)
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
def test_model_and_clusters_pickle_roundtrip_hip(hip_lib):
    a, b = _abi.Hamsm(lib=hip_lib), _abi.Hamsm(lib=hip_lib)
    try:
        _roundtrip(a, b)
    finally:
        a.close()
        b.close()


# ---- B1 --------------------------------------------------------------------------------------------------
class _StubWorkManager:
    def __init__(self, master=True):
        self.is_master = master


class _StubDataManager:
    we_h5filename = "west.h5"

    def __init__(self):
        self.closed = 0
        self.hamsm_model = "unset"

    def close_backing(self):
        self.closed += 1


class _StubSimManager:
    def __init__(self, master=True, bin_mapper=None):
        self.work_manager = _StubWorkManager(master)
        self.data_manager = _StubDataManager()
        self.callbacks = []
        self.system = type("System", (), {"bin_mapper": bin_mapper})()

    def finalize_run(self):
        for _, fn in sorted(self.callbacks, key=lambda x: x[0]):
            fn()

    def register_callback(self, hook, fn, priority):
        assert hook == self.finalize_run
        self.callbacks.append((priority, fn))


def flatten_featurizer(self, coords):
    return coords.reshape(coords.shape[0], -1)


def _plugin_case(h, monkeypatch):
    from msm_we_amd.data import InMemorySource
    from msm_we_amd.westpa_plugins.hamsm_driver import HAMSMDriver

    monkeypatch.setenv("OMP_NUM_THREADS", "1")
    g, meta = load_golden("c1p_d10")
    bm = RectilinearBinMapper([np.array(meta["boundaries"], dtype=np.float32)])
    sm = _StubSimManager(bin_mapper=bm)
    cfg = dict(model_name="plugin-test", n_clusters=meta["K_s"], tau=meta["tau"], basis_pcoord_bounds=meta["basis"],
               target_pcoord_bounds=meta["target"], dimreduce_method="none", featurization=f"{__name__}.flatten_featurizer",
               iteration_source=InMemorySource(golden_iterations(g)), random_state=meta["seed"], cross_validation_groups=2,
               n_cpus=4, ray_kwargs={"ignored": True})
    # not the master: nothing is registered (hamsm_driver.py:43-45)
    idle = HAMSMDriver(_StubSimManager(master=False), cfg)
    assert not hasattr(idle, "sim_manager")
    drv = HAMSMDriver(sm, cfg)
    assert drv.priority == 2 and drv.h5file_paths == ["west.h5"] and drv.first_iter_to_use == 1
    assert sm.callbacks == [(2, drv.construct_hamsm)]
    # route the model to the test's context
    import msm_we_amd.model as mm

    orig_init = mm.modelWE.__init__
    monkeypatch.setattr(mm.modelWE, "__init__", lambda self, h_=None, device=0: orig_init(self, h=h, device=device))
    sm.finalize_run()
    model = sm.data_manager.hamsm_model
    assert sm.data_manager.closed == 1 and model is not None and model.modelName == "plugin-test"
    assert sm.data_manager.processCoordinates is flatten_featurizer
    # the same observables as the reference's run of this input (golden c1p_d10), through the plugin entry point;
    # k-means++ exact ties are not teacher-forced here, so compare what does not depend on the one flipped seed
    assert model.n_clusters + 2 == model.fluxMatrix.shape[0] == model.Tmatrix.shape[0]
    assert abs(model.fluxMatrix.sum() - 1.0) < 1e-12 and abs(model.pSS.sum() - 1.0) < 1e-9
    assert np.array_equal(np.concatenate(model.post_cluster_model.dtrajs), g["dtrajs_post_cluster"])
    assert np.array_equal(model.fluxMatrix, g["fluxMatrix"])
    assert abs(model.JtargetSS / float(g["JtargetSS"]) - 1) < 1e-6
    assert len(model.validation_models) == 2 and all(v.JtargetSS > 0 for v in model.validation_models)
    # what the later plugins do with this object (restart_driver.py:1142: `pickle.dump(model, ...)`): the plugin's model is an
    # importable class that carries its featurizer by NAME
    assert type(model) is mm.modelWE and model._featurizer_spec == f"{__name__}.flatten_featurizer"
    blob = pickle.dumps(model, protocol=4)
    assert b"flatten_featurizer" in blob
    m2 = pickle.loads(blob)
    assert type(m2) is mm.modelWE and m2._featurizer is None and m2._h is None
    xyz = np.arange(24.0).reshape(2, 4, 3)
    assert np.array_equal(m2.processCoordinates(xyz), xyz.reshape(2, 12)) and m2._featurizer is flatten_featurizer
    assert np.array_equal(m2.fluxMatrix, model.fluxMatrix) and np.array_equal(m2.pSS, model.pSS)
    m2.attach(h)
    it = 5
    m2.load_iter_data(it)
    assert np.array_equal(m2.clusters.predict(golden_iterations(g)[it]["Xc"]), model.dtrajs[it - 1])
    # a featurizer given as an object (not a path) is carried as that object: picklable iff the function is importable
    m3 = mm.modelWE(h)
    m3.set_featurizer(flatten_featurizer)
    assert pickle.loads(pickle.dumps(m3))._featurizer is flatten_featurizer


def test_westpa_plugin_shim_oracle(oracle_h, monkeypatch):
    _plugin_case(oracle_h, monkeypatch)


@pytest.mark.gpu
def test_westpa_plugin_shim_hip(hip_h, monkeypatch):
    _plugin_case(hip_h, monkeypatch)

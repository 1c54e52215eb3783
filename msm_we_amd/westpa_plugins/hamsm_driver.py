"""``HAMSMDriver``: the WESTPA plugin that builds an haMSM when a run finishes -- level B1 of the drop-in boundary
(SURVEY.md §8b).  Mirrors ``msm_we/westpa_plugins/hamsm_driver.py:8-144``: same constructor contract
(``__init__(sim_manager, plugin_config)``), same ``plugin_config`` keys and defaults (``:59-80``, ``:88-101``), registered on
``sim_manager.finalize_run`` at ``priority`` (default 2) on the master only (``:44-57``), result stored as
``data_manager.hamsm_model`` (``:141-142``).  The model it builds is :class:`msm_we_amd.model.modelWE`, i.e. the hot
path runs in the HIP library; nothing here touches a GPU until ``construct_hamsm`` is called.

west.cfg::

    west:
      plugins:
        - plugin: msm_we_amd.westpa_plugins.hamsm_driver.HAMSMDriver
          model_name: ...            n_clusters: ...         tau: ...
          basis_pcoord_bounds: [[lo, hi], ...]               target_pcoord_bounds: [[lo, hi], ...]
          first_analysis_iter: 1     dimreduce_method: none  featurization: package.module.function
          # additions of this package (the reference reads both from the WESTPA run through westpa / h5py):
          user_bin_mapper: package.module.object   # default: sim_manager.system.bin_mapper
          iteration_source: package.module.object  # default: WestpaH5Source(we_h5filename) (needs h5py)

Differences, all forced by the environment: ``westpa`` is imported lazily (``westpa.rc.pstatus`` falls back to logging,
``extloader.get_object`` to importlib) so the class is importable and testable where WESTPA is not installed; Ray options
(``n_cpus``, ``ray_address``, ``ray_kwargs``) are accepted and ignored -- there is no Ray.
"""
from __future__ import annotations

import logging
import os

log = logging.getLogger("msm_we_amd")

__all__ = ["HAMSMDriver"]


def _pstatus(msg):
    try:
        import westpa

        westpa.rc.pstatus(msg)
    except Exception:  # noqa: BLE001 - westpa absent or not configured
        log.info(msg)


def _get_object(spec):
    """``westpa.core.extloader.get_object``: 'pkg.module.attr' -> object (objects pass through)."""
    from ..data import resolve_object

    return resolve_object(spec)


class HAMSMDriver:
    """WESTPA plugin to construct an haMSM (reference ``hamsm_driver.py:8-144``)."""

    def __init__(self, sim_manager, plugin_config):
        _pstatus("Initializing haMSM plugin")
        # hamsm_driver.py:39-41 asserts sklearn's effective OpenMP thread count; the host side here uses BLAS only in the
        # solve and bounds its pool itself, so the environment variable is what is checked
        assert os.environ.get("OMP_NUM_THREADS", "1") == "1", "Set $OMP_NUM_THREADS=1 for proper msm-we functionality"
        if not sim_manager.work_manager.is_master:  # :43-45
            _pstatus("Not running on the master process, skipping")
            return
        self.data_manager = sim_manager.data_manager
        self.sim_manager = sim_manager
        self.plugin_config = plugin_config
        self.priority = plugin_config.get("priority", 2)  # :52-53
        sim_manager.register_callback(sim_manager.finalize_run, self.construct_hamsm, self.priority)  # :55-57
        self.h5file_paths = [self.data_manager.we_h5filename]  # :64 (extended by the restart plugin)
        self.first_iter_to_use = plugin_config.get("first_analysis_iter", 1)
        self.dimreduce_use_weights = plugin_config.get("dimreduce_use_weights", True)
        self.dimreduce_var_cutoff = plugin_config.get("dimreduce_var_cutoff", None)
        self.cross_validation_groups = plugin_config.get("cross_validation_groups", 2)
        self.ray_address = plugin_config.get("ray_address", None)  # accepted, unused
        self.ray_kwargs = plugin_config.get("ray_kwargs", {})
        self.device = plugin_config.get("device", 0)

    def _iteration_source(self):
        from ..data import WestpaH5Source

        src = self.plugin_config.get("iteration_source")
        if src is not None:
            src = _get_object(src)
            return src(self.h5file_paths) if callable(src) and not hasattr(src, "iterations") else src
        pcoord_ndim = len(self.plugin_config.get("basis_pcoord_bounds") or [[0, 0]])
        return WestpaH5Source(self.h5file_paths[0], auxpath=self.plugin_config.get("auxpath", "coord"), pcoord_ndim=pcoord_ndim)

    def _bin_mapper(self):
        bm = self.plugin_config.get("user_bin_mapper")
        if bm is not None:
            return _get_object(bm)
        system = getattr(self.sim_manager, "system", None)
        bm = getattr(system, "bin_mapper", None)
        if bm is None:
            raise ValueError("no bin mapper: give `user_bin_mapper` in the plugin config (sim_manager.system.bin_mapper is absent)")
        return bm

    def construct_hamsm(self):
        """Build an haMSM for later plugins; stored on the data manager (``hamsm_driver.py:82-144``)."""
        from ..model import modelWE

        cfg = self.plugin_config
        self.data_manager.hamsm_model = None  # :87
        featurizer = _get_object(cfg.get("featurization"))  # :99-103
        self.data_manager.close_backing()  # :105
        model = modelWE(device=self.device)
        if featurizer is not None:
            # the reference patches the CLASS (`msm_we.modelWE.processCoordinates = featurizer`, :101); here the featurizer is
            # bound to this model alone, by its dotted path where the config gave one, so that the model the later plugins
            # pickle (restart_driver.py:1142) is an importable class carrying a name, not a function object
            model.set_featurizer(cfg.get("featurization"))
            self.data_manager.processCoordinates = featurizer
        model.set_iteration_source(self._iteration_source())
        pcoord_ndim = len(cfg.get("basis_pcoord_bounds"))
        model.build_analyze_model(
            file_paths=self.h5file_paths,
            ref_struct=cfg.get("ref_pdb_file"),
            modelName=cfg.get("model_name"),
            basis_pcoord_bounds=cfg.get("basis_pcoord_bounds"),
            target_pcoord_bounds=cfg.get("target_pcoord_bounds"),
            dimreduce_method=cfg.get("dimreduce_method", None),
            n_clusters=cfg.get("n_clusters"),
            tau=cfg.get("tau", None),
            ray_kwargs={"num_cpus": cfg.get("num_cpus", None), **self.ray_kwargs},
            step_kwargs={
                "initialize": {"pcoord_ndim": pcoord_ndim},
                "dimReduce": {"use_weights": self.dimreduce_use_weights, "variance_cutoff": self.dimreduce_var_cutoff,
                              "first_iter": self.first_iter_to_use, "first_rough_iter": self.first_iter_to_use},
                "clustering": {"first_cluster_iter": self.first_iter_to_use, "user_bin_mapper": self._bin_mapper(),
                               **({"random_state": cfg["random_state"]} if "random_state" in cfg else {})},
            },
            fluxmatrix_iters=[self.first_iter_to_use, -1],
            allow_validation_failure=True,  # :135
            cross_validation_groups=self.cross_validation_groups,
        )
        release = getattr(getattr(model, "_source", None), "close", None)
        if callable(release):
            release()  # no handle on west.h5 outlives the build: WESTPA / the restart driver reopen the file for writing
        _pstatus(f"Storing built haMSM on {self.data_manager}")
        self.data_manager.hamsm_model = model  # :141-142
        return model

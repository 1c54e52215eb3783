"""Synthetic WE-shaped input generator.

The same code makes the tiny golden inputs (tests/golden/make_golden.py, in the
build container) and the large benchmark tensors on the GPU box, so parity
fixtures and bench workloads are drawn from one distribution.

One WE iteration = ``segs`` segments.  Each segment has a parent frame and a
child frame (feature rows, stored float32 like WESTPA's ``auxdata/coord``,
reference ``msm_we/westpa_plugins/augmentation_driver.py:173-180``), a parent
and a child progress coordinate (float64, 1-D) and a weight (float64, sums to 1
per iteration), i.e. exactly what ``DataMixin.load_iter_data`` /
``get_transition_data_lag0`` hand the hot path (``_data.py:807-932,254-320``).

Iterations are seeded independently (``default_rng([seed, iteration])``) so a
rank can generate only its own shard of iterations.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

__all__ = ["WEConfig", "CONFIGS", "latent_centres", "make_iteration", "make_block",
           "strata_boundaries", "basis_target_bounds", "stage_streamed"]


@dataclass(frozen=True)
class WEConfig:
    name: str
    seed: int
    n_iters: int          # iterations 1..n_iters carry data
    segs_per_iter: int
    D: int                # features per frame
    S: int                # strata (WE bins)
    K_s: int              # clusters per stratum

    @property
    def n_frames(self) -> int:
        return self.n_iters * self.segs_per_iter

    @property
    def K(self) -> int:
        return self.S * self.K_s


# C1' replaces the unavailable NTL9 run; C2..C5 are BASELINE.json configs[1..4].
CONFIGS = {
    "c1p_d10": WEConfig("c1p_d10", 1337, 40, 128, 10, 2, 25),
    "c1p_d30": WEConfig("c1p_d30", 1337, 40, 128, 30, 2, 25),
    "c2": WEConfig("c2", 2, 1000, 1000, 10, 1, 200),
    "c3": WEConfig("c3", 3, 1000, 10000, 30, 4, 250),
    "c5": WEConfig("c5", 5, 1000, 50000, 64, 4, 1250),
}


def strata_boundaries(S: int) -> np.ndarray:
    """float32 rectilinear boundaries: unit intervals [s, s+1), the last open to +inf."""
    b = np.arange(S + 1, dtype=np.float32)
    b[-1] = np.inf
    return b


def basis_target_bounds(S: int):
    """basis = (-inf, 0.05), target = (S-0.05, +inf); strict inequalities (msm_we.py:482-488)."""
    return np.array([[-np.inf, 0.05]]), np.array([[S - 0.05, np.inf]])


def latent_centres(cfg: WEConfig) -> np.ndarray:
    rng = np.random.default_rng([cfg.seed, 0xC0FFEE])
    return (rng.standard_normal((cfg.S * cfg.K_s, cfg.D)) * 4.0).astype(np.float32)


def _frames(rng, p, cfg: WEConfig, latent: np.ndarray) -> np.ndarray:
    s = np.minimum(np.floor(p[:, 0]).astype(np.int64), cfg.S - 1)
    which = rng.integers(0, cfg.K_s, size=p.shape[0])
    X = latent[s * cfg.K_s + which]
    X = X + rng.standard_normal(X.shape, dtype=np.float32)
    return np.ascontiguousarray(X, dtype=np.float32)


def make_iteration(cfg: WEConfig, it: int, latent: np.ndarray | None = None, segs: int | None = None):
    """Data of WE iteration ``it`` (1-based): dict(Xp, Xc f32[n,D]; p0, p1 f64[n,1]; w f64[n])."""
    if latent is None:
        latent = latent_centres(cfg)
    n = cfg.segs_per_iter if segs is None else segs
    rng = np.random.default_rng([cfg.seed, int(it)])
    p0 = rng.uniform(0.0, cfg.S, size=(n, 1))
    p1 = np.clip(p0 + rng.normal(0.0, 0.35, size=(n, 1)), 0.0, cfg.S + 0.5)
    Xp = _frames(rng, p0, cfg, latent)
    Xc = _frames(rng, p1, cfg, latent)
    w = rng.exponential(1.0, size=n)
    w /= w.sum()
    return {"Xp": Xp, "Xc": Xc, "p0": p0, "p1": p1, "w": w}


def make_block(cfg: WEConfig, it_first: int, it_last: int, latent: np.ndarray | None = None):
    """Iterations it_first..it_last (inclusive) concatenated, plus ``offsets`` (CSR-style) per iteration."""
    if latent is None:
        latent = latent_centres(cfg)
    parts = [make_iteration(cfg, it, latent) for it in range(it_first, it_last + 1)]
    out = {k: np.concatenate([p[k] for p in parts], axis=0) for k in ("Xp", "Xc", "p0", "p1", "w")}
    sizes = np.array([p["w"].shape[0] for p in parts], dtype=np.int64)
    out["offsets"] = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    out["iters"] = np.arange(it_first, it_last + 1, dtype=np.int64)
    return out


def stage_streamed(h, cfg: WEConfig, it_first: int, n_iters: int, log=None, slab_iters: int = 16, threads: int | None = None,
                   keep=()):
    """Iterations it_first .. it_first+n_iters-1 staged into HBM without ever being whole on the host: they are generated
    ``slab_iters`` at a time (by a thread pool: the NumPy generators release the GIL) and every slab is uploaded into its place
    in device arrays allocated once.  BASELINE config 5 is 2 x 12.8 GB of float32 frames: the host side stays at one slab.
    Returns ``(DeviceStore, offsets)`` -- and, for the keys named in ``keep`` (of "p0", "p1", "w"), the host copies too:
    ``(DeviceStore, offsets, {key: array})``."""
    import os
    import time
    from concurrent.futures import ThreadPoolExecutor

    from . import _abi
    from .data import DeviceStore

    lat = latent_centres(cfg)
    seg, D = cfg.segs_per_iter, cfg.D
    n = n_iters * seg
    dXp, dXc = h.alloc((n, D), np.float32), h.alloc((n, D), np.float32)
    dp0, dp1, dw = h.alloc((n, 1), np.float64), h.alloc((n, 1), np.float64), h.alloc(n, np.float64)
    kept = {k: np.empty((n, 1) if k != "w" else n) for k in keep}
    threads = threads or min(16, os.cpu_count() or 1)
    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        def submit(k0):
            return [ex.submit(make_iteration, cfg, it, lat) for it in range(it_first + k0, it_first + min(n_iters, k0 + slab_iters))]

        nxt = submit(0)
        for k0 in range(0, n_iters, slab_iters):
            k1 = min(n_iters, k0 + slab_iters)
            parts = [f.result() for f in nxt]
            nxt = submit(k1) if k1 < n_iters else []   # the next slab is generated while this one is uploaded
            a = k0 * seg
            for key, dst, width in (("Xp", dXp, D), ("Xc", dXc, D), ("p0", dp0, 1), ("p1", dp1, 1), ("w", dw, 1)):
                slab = np.concatenate([p[key] for p in parts], axis=0)
                h.h2d(dst.offset(a * width), slab)
                if key in kept:
                    kept[key][a: a + len(slab)] = slab
            if log and (k0 // slab_iters) % 8 == 0:
                log(f"generated + uploaded {k1}/{n_iters} iterations ({time.time() - t0:.1f}s)")
    offs = np.arange(n_iters + 1, dtype=np.int64) * seg
    store = DeviceStore(h, D, 1, _abi.F32, offs, dXp, dXc, dp0, dp1, dw)
    return (store, offs, kept) if keep else (store, offs)

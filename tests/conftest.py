"""Shared fixtures.  `-m "not gpu"`: oracle vs golden vectors, host logic, ABI export checks.
`-m gpu`: parity tests proper, HIP library vs oracle, all through the C ABI."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "experiments: the training experiments that measured slower than the shipped path and are not "
                                       "part of libhamsm_hip.so (csrc/experiments/, DESIGN.md 5.3a-d): `make -C msm_we_amd/csrc "
                                       "EXPERIMENTS=1`, then `pytest -m experiments` on a GPU box")


def pytest_collection_modifyitems(config, items):
    """tests of csrc/experiments/ run only when asked for by their own marker (`-m experiments`): `-m gpu` skips them"""
    if "experiments" in (config.option.markexpr or ""):
        return
    skip = pytest.mark.skip(reason="experiment, not part of the shipped library: make EXPERIMENTS=1 && pytest -m experiments")
    for it in items:
        if it.get_closest_marker("experiments"):
            it.add_marker(skip)


FORCED_SEEDINGS = {}   # golden case -> (k-means++ seedings replaced by the reference's, seedings compared)


def pytest_terminal_summary(terminalreporter):
    """the teacher-forced k-means++ seedings (exact mathematical ties that the reference breaks by BLAS
    rounding) are asserted per test AND reported here, per golden case."""
    if FORCED_SEEDINGS:
        terminalreporter.write_line("k-means++ seedings teacher-forced (exact ties) / compared, per golden case: " +
                                    ", ".join(f"{k}: {a}/{b}" for k, (a, b) in sorted(FORCED_SEEDINGS.items())))
    if TIE_WITNESSES:
        terminalreporter.write_line("tie witnesses (seeding, greedy step, exact relative difference of the two potentials): " +
                                    ", ".join(f"{k}: " + " ".join(f"({i}, {s}, {r:.1e})" for i, s, r in v)
                                              for k, v in sorted(TIE_WITNESSES.items())))


def _build_oracle():
    if os.environ.get("HAMSM_ORACLE_SO"):   # oracle/run_asan.sh: the sanitizer build of the same source
        return os.environ["HAMSM_ORACLE_SO"]
    so = os.path.join(ROOT, "oracle", "_build", "libhamsm_oracle.so")
    src = os.path.join(ROOT, "oracle", "hamsm_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return so


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure): same C ABI as the product, plain C."""
    return ctypes.CDLL(_build_oracle())


@pytest.fixture()
def oracle_h(oracle_lib):
    from msm_we_amd import _abi

    h = _abi.Hamsm(lib=oracle_lib)
    yield h
    h.close()


@pytest.fixture(scope="session")
def hip_lib():
    from msm_we_amd import _abi

    if os.environ.get("HAMSM_HIP_STANDIN"):
        # sanitizer runs only (oracle/run_asan.sh --gpu-logic): the GPU suite's host logic with a second copy of the oracle in
        # the product's seat, so that AddressSanitizer sees every buffer the tests hand across the ABI.  Never set by the driver.
        # the switch must not be able to turn `-m gpu` green on a box that HAS a GPU -- there it is an error.
        if os.path.exists("/dev/kfd"):
            raise RuntimeError("HAMSM_HIP_STANDIN is set on a machine with a GPU (/dev/kfd): the stand-in is for the CPU "
                               "sanitizer script only (oracle/run_asan.sh --gpu-logic); unset it")
        import shutil, tempfile
        twin = os.path.join(tempfile.mkdtemp(), "libhamsm_standin.so")
        shutil.copy(os.environ["HAMSM_HIP_STANDIN"], twin)
        return ctypes.CDLL(twin)
    return _abi.load()  # raises if the HIP library is missing: no fallback


@pytest.fixture(scope="session")
def exp_lib():
    """libhamsm_hip_exp.so: the product's sources + csrc/experiments/ (make EXPERIMENTS=1); never loaded by the product"""
    path = os.path.join(ROOT, "msm_we_amd", "csrc", "libhamsm_hip_exp.so")
    if not os.path.exists(path):
        pytest.skip("libhamsm_hip_exp.so not built (make -C msm_we_amd/csrc EXPERIMENTS=1)")
    return ctypes.CDLL(path)


@pytest.fixture()
def hip_h(hip_lib):
    from msm_we_amd import _abi

    h = _abi.Hamsm(lib=hip_lib, device=0)
    assert h.backend == "hip-gfx950" or os.environ.get("HAMSM_HIP_STANDIN")
    yield h
    h.close()


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def any_h(request):
    """the same test body on the oracle (CPU suite) and, under `-m gpu`, on the HIP library (host-side
    pipelines -- staging, the NTL9 fixture chain -- were only ever run on the oracle's handle)"""
    return request.getfixturevalue("oracle_h" if request.param == "oracle" else "hip_h")


class RectilinearBinMapper:
    """Test-side stand-in for westpa.core.binning.RectilinearBinMapper (attributes only: the
    library does the assignment)."""

    def __init__(self, boundaries):
        self.boundaries = [np.asarray(b, dtype=np.float32) for b in boundaries]
        self.nbins = int(np.prod([len(b) - 1 for b in self.boundaries]))


class _Golden(dict):
    """a golden case as a dict of arrays with np.load's `.files` (compact cases are expanded on load)"""

    @property
    def files(self):
        return list(self.keys())


def load_golden(case):
    z = np.load(os.path.join(GOLDEN, f"{case}.npz"))
    meta = json.loads(str(z["meta"]))
    if "parents_clean" not in z.files:
        return z, meta
    # compact case (make_golden.compact): inputs are regenerated from the seed with the generator that produced them,
    # COO triplets become dense matrices again, (parent, child) pairs are re-joined, T is left to the caller
    from msm_we_amd import synthetic

    g = _Golden()
    for k in z.files:
        if k.endswith("__coo_rc"):
            base = k[: -len("__coo_rc")]
            n = int(z[base + "__coo_n"])
            M = np.zeros((n, n))
            rc = z[k]
            M[rc[0], rc[1]] = z[base + "__coo_v"]
            g[base] = M
        elif k.endswith("__coo_v") or k.endswith("__coo_n"):
            continue
        else:
            v = z[k]
            g[k] = v.astype(np.int64) if v.dtype == np.int16 else v
    cfg = synthetic.WEConfig(meta["case"], meta["gen_seed"], meta["n_iters"], meta["segs_per_iter"], meta["D"], meta["S"], meta["K_s"])
    blk = synthetic.make_block(cfg, 1, cfg.n_iters)
    for k in ("Xp", "Xc", "p0", "p1", "w"):
        g["in_" + k] = blk[k]
    assert np.array_equal(blk["offsets"], g["in_offsets"])
    for tag in ("post_cluster", "clean"):
        g[f"pairs_{tag}"] = np.stack([g[f"parents_{tag}"], g[f"dtrajs_{tag}"]], axis=1)
    return g, meta


def check_committor(m, g):
    """`get_committor` (_analysis.py:527-606; dense matmul iteration in the reference, CSR mat-vec here) against the reference's
    vector: same iteration count by construction of the stopping rule, values to 1e-12"""
    if "committor" not in g.files:
        return
    m.get_committor()
    assert m.q.shape == g["committor"].shape
    np.testing.assert_allclose(m.q, g["committor"], rtol=1e-12, atol=1e-15)


def check_cluster_structures(m, g):
    """f4: `update_cluster_structures` on the organised model against the reference's: clusters in the order the reference
    first meets them, per cluster the same segments (iteration, segment), weights, pcoords and coordinates -- or the same
    failure where the reference fails (NaN coordinates)."""
    if "ucs_error" in g.files:
        with pytest.raises(Exception) as ei:
            m.update_cluster_structures(build_pcoord_cache=True)
        assert type(ei.value).__name__ == str(g["ucs_error"]).split(":")[0]
        return
    m.update_cluster_structures(build_pcoord_cache=True)
    keys = list(m.cluster_structures.keys())
    assert keys == list(g["ucs_keys"])
    assert [len(m.cluster_structures[k]) for k in keys] == list(g["ucs_counts"])
    if "ucs_weights" in g.files:   # (compact cases keep the per-cluster summaries only)
        assert np.array_equal(np.concatenate([np.asarray(m.cluster_structure_weights[k]) for k in keys]), g["ucs_weights"])
        its = np.concatenate([np.array([[a, b] for a, b, _ in m.structure_iteration_segments[k]]) for k in keys])
        assert np.array_equal(its, g["ucs_iter_seg"])
    assert np.array_equal(np.stack([np.asarray(m.cluster_structures[k][0], dtype=np.float64).ravel() for k in keys]), g["ucs_first_coord"])
    np.testing.assert_allclose([np.sum(np.asarray(m.cluster_structures[k], dtype=np.float64)) for k in keys], g["ucs_coord_sums"], rtol=1e-13)
    if "ucs_pcoords" in g.files:
        pcs = np.concatenate([np.asarray(m.pcoord_cache[k], dtype=np.float64).reshape(len(m.pcoord_cache[k]), -1) for k in keys])
        assert np.array_equal(pcs, g["ucs_pcoords"])


def check_block_validation(m, g):
    """a14: `do_block_validation` (msm_we.py:884-1009) against what the REAL reference produced for the same split:
    iteration lists, per-group raw and organised flux matrices, cleaned state counts, discretization and remap identical;
    steady state and target flux within 1e-6 rel."""
    if not bool(g["bv_ok"]):
        with pytest.raises(Exception):
            m.do_block_validation(2, 4)
        return
    groups, blocks = (int(x) for x in g["bv_groups_blocks"])
    m.do_block_validation(cross_validation_groups=groups, cross_validation_blocks=blocks)
    assert len(m.validation_models) == groups
    for gi, (its, vm) in enumerate(zip(m.validation_iterations, m.validation_models)):
        assert list(its) == list(g[f"bv_iters_{gi}"])
        assert vm.n_clusters == int(g[f"bv_n_clusters_{gi}"])
        assert [vm.clusters.we_remap[b] for b in range(len(g[f"bv_remap_{gi}"]))] == list(g[f"bv_remap_{gi}"])
        assert np.array_equal(np.concatenate(vm.dtrajs), g[f"bv_dtrajs_{gi}"])
        assert np.array_equal(vm.fluxMatrixRaw, g[f"bv_fluxMatrixRaw_{gi}"])
        assert np.array_equal(vm.fluxMatrix, g[f"bv_fluxMatrix_{gi}"])
        np.testing.assert_allclose(vm.pSS, g[f"bv_pSS_{gi}"], rtol=1e-6, atol=1e-300)
        assert abs(vm.JtargetSS - float(g[f"bv_JtargetSS_{gi}"])) <= 1e-6 * float(g[f"bv_JtargetSS_{gi}"])


def golden_iterations(g):
    offs = g["in_offsets"]
    return {i + 1: {k: g["in_" + k][offs[i]: offs[i + 1]] for k in ("Xp", "Xc", "p0", "p1", "w")}
            for i in range(len(offs) - 1)}


def _exact_ints(A):
    """a float64 array as exact Python integers times 2**emin (object array, emin): every float64 is m * 2**e"""
    A = np.asarray(A, dtype=np.float64)
    if not np.all(np.isfinite(A)):
        raise ValueError("tie witness: non-finite data")
    m, e = np.frexp(A)
    mi = (m * 2.0 ** 53).astype(np.int64)          # exact: 53-bit significands
    ee = e.astype(np.int64) - 53
    nz = A != 0
    emin = int(ee[nz].min()) if nz.any() else 0
    out = np.zeros(A.shape, dtype=object)
    flat_o, flat_m, flat_e, flat_nz = out.reshape(-1), mi.reshape(-1), ee.reshape(-1), nz.reshape(-1)
    for i in range(flat_o.size):
        flat_o[i] = (int(flat_m[i]) << int(flat_e[i] - emin)) if flat_nz[i] else 0
    return out, emin


def tie_witness(Xb, wb, mine, ref):
    """a k-means++ seeding that differs from the reference's may only be teacher-forced if it is a TIE.
    At the first greedy step where the two index lists diverge both runs hold the same centres and the same candidates
    (sklearn `_kmeans.py:214-276`) and picked different ones; the step minimises pot(c) = sum_i w_i min(closest_i, |x_i - x_c|^2).
    Both potentials are evaluated here in EXACT integer arithmetic on the float64 data; they must agree to 4 ulp (2^-50
    relative) -- i.e. no float64 summation order can be said to be the right one -- or the test fails: the difference would
    then be this build's own arithmetic, not a tie.  Returns (step, relative difference as a float)."""
    mine, ref = np.asarray(mine), np.asarray(ref)
    c = int(np.nonzero(mine != ref)[0][0])
    assert c >= 1, "the first centre is drawn by the host (random_state.choice): it cannot differ"
    Xo, _ = _exact_ints(Xb)
    wo = np.ones(len(Xb), dtype=object) if wb is None else _exact_ints(wb)[0]
    closest = None
    for j in ref[:c]:
        d = ((Xo - Xo[int(j)]) ** 2).sum(axis=1)
        closest = d if closest is None else np.minimum(closest, d)
    pots = []
    for cand in (int(mine[c]), int(ref[c])):
        d = ((Xo - Xo[cand]) ** 2).sum(axis=1)
        pots.append(int((wo * np.minimum(closest, d)).sum()))
    pa, pb = pots
    big = max(abs(pa), abs(pb), 1)
    assert abs(pa - pb) << 50 <= big, (f"seeding differs from the reference's at greedy step {c} and it is NOT a tie: exact potentials "
                                       f"differ by {abs(pa - pb) / big:.3e} relative (candidates {int(mine[c])} / {int(ref[c])})")
    return c, abs(pa - pb) / big


TIE_WITNESSES = {}   # golden case -> [(stratum seeding number, greedy step, exact relative difference of the two potentials)]
# The seedings that ARE teacher-forced, per case: pinned.  Each is a witnessed exact tie (`tie_witness`: the two candidates'
# potentials are equal in exact arithmetic -- two mutually nearest isolated rows -- and the reference breaks it by BLAS rounding,
# the oracle by its own summation order); every other seeding of every case must equal the reference's index for index.
EXPECTED_FORCED = {"c1p_d30": 1, "grid2d_d10": 1, "late_start_d10": 1, "fuzz_03": 2, "fuzz_53": 1}


def expected_forced(case):
    return EXPECTED_FORCED.get(case, 0)



def build_model(h, g, meta, seed_forcing=True, planned=True, source=None, mapper_dfunc=None):
    """modelWE on golden inputs, up to (and including) cluster_coordinates.  Returns (model, state); state["forced"] counts
    k-means++ seedings replaced by the reference's -- only ever after `tie_witness` has shown, in exact arithmetic, that the
    reference's choice and this build's are a tie at the step where they diverge."""
    from msm_we_amd import _abi
    from msm_we_amd.data import InMemorySource
    from msm_we_amd.model import modelWE

    m = modelWE(h=h)
    m.set_iteration_source(source if source is not None else InMemorySource(golden_iterations(g)))
    m.initialize(["golden.h5"], {"coords": None, "nAtoms": meta["D"], "coord_ndim": 1}, "golden",
                 basis_pcoord_bounds=meta["basis"], target_pcoord_bounds=meta["target"], tau=meta["tau"],
                 pcoord_ndim=meta.get("pcoord_ndim", 1), use_weights_in_clustering=meta["use_weights"],
                 dim_reduce_method=meta.get("dim_reduce", "none"))
    m.get_iterations()
    m.get_coordSet(m.maxIter)
    m.dimReduce()
    state = {"i": 0, "forced": 0, "chosen": [], "witness": []}
    orig = h.kmeans_seed

    def batch_on_host(d_X, x_dtype, ldx, x_off, d_rows, rows_off, n_b, d_w, sub):
        """the seeding's batch as k-means++ sees it: float64 rows (and weights) in batch / subsample order"""
        dt = np.float32 if int(x_dtype) == _abi.F32 else np.float64
        rows = np.empty(int(n_b), dtype=np.int32)
        h.d2h(rows, _abi._dp(d_rows) + 4 * int(rows_off))
        span = int(rows.max()) + 1
        X = np.empty((span, int(ldx)), dtype=dt)
        h.d2h(X, _abi._dp(d_X) + int(x_off) * int(ldx) * dt().itemsize)
        Xb = X[rows][:, : h.D].astype(np.float64)
        wb = None
        if d_w is not None:
            w = np.empty(span, dtype=np.float64)
            h.d2h(w, _abi._dp(d_w) + 8 * int(x_off))
            wb = w[rows]
        if sub is not None:
            Xb, wb = Xb[np.asarray(sub)], (None if wb is None else wb[np.asarray(sub)])
        return Xb, wb

    def settle(stratum, ch, batch):
        """compare one seeding with the reference's indices; a difference must be a witnessed tie, and is then teacher-forced"""
        ref = g["pp_indices"][state["i"]]
        state["chosen"].append((np.asarray(ch).copy(), ref.copy()))
        if seed_forcing and not np.array_equal(ch, ref):
            step, rel = tie_witness(*batch(), ch, ref)
            state["witness"].append((state["i"], step, rel))
            state["forced"] += 1
            h.set_centroids(stratum, g["pp_centers"][state["i"]])
        state["i"] += 1

    def seed(stratum, d_X, x_dtype, ldx, d_rows, n_b, d_w, sub, K_s, first, u, L, want_chosen=False):
        ch = orig(stratum, d_X, x_dtype, ldx, d_rows, n_b, d_w, sub, K_s, first, u, L, want_chosen=True)
        settle(stratum, ch, lambda: batch_on_host(d_X, x_dtype, ldx, 0, d_rows, 0, n_b, d_w, sub))
        return ch

    orig_multi = h.kmeans_seed_multi

    def seed_multi(strata, d_X, x_dtype, ldx, x_off, d_rows, rows_off, n_b, d_w, subs, K, first, us, L):
        # the planned training path seeds several strata in one call (same order as the reference's k-means++ calls): the
        # CHOSEN INDICES of every item are compared (hamsm_kmeans_seed_chosen, ABI 7) -- not the centroids' bits, which differ
        # by an ulp wherever the features do (PCA golden) although the seeding is the same
        orig_multi(strata, d_X, x_dtype, ldx, x_off, d_rows, rows_off, n_b, d_w, subs, K, first, us, L)
        chosen = [h.kmeans_seed_chosen(i, K[i]) for i in range(len(strata))]   # (all read before a forced set_centroids)
        for i, s in enumerate(strata):
            settle(s, chosen[i], lambda i=i: batch_on_host(d_X, x_dtype, ldx, x_off[i], d_rows, rows_off[i], n_b[i], d_w, subs[i]))

    h.kmeans_seed = seed
    h.kmeans_seed_multi = seed_multi
    m.speculative_training = planned
    bm = RectilinearBinMapper([np.array(meta["boundaries"], dtype=np.float32)])
    if meta.get("grid2d"):
        bm = RectilinearBinMapper([np.array(b, dtype=np.float32) for b in meta["boundaries_nd"]])
    if meta.get("voronoi"):
        from msm_we_amd.binning import VoronoiBinMapper, euclidean_dfunc

        # mapper_dfunc = None: the tagged Euclidean function (bins computed on the device); anything else: the mapper's own
        # `assign` on the host (external path)
        bm = VoronoiBinMapper(mapper_dfunc or euclidean_dfunc, np.array(meta["voronoi"], dtype=np.float32))
    try:
        m.cluster_coordinates(meta["K_s"], streaming=True, use_ray=True, stratified=True, user_bin_mapper=bm,
                              random_state=meta["seed"], store_validation_model=True,
                              **({"first_cluster_iter": meta["first_iter"]} if meta.get("first_iter", 1) != 1 else {}))
    finally:
        h.kmeans_seed = orig
        h.kmeans_seed_multi = orig_multi
    if seed_forcing:
        prev = FORCED_SEEDINGS.get(meta["case"], (0, 0))
        FORCED_SEEDINGS[meta["case"]] = (max(prev[0], state["forced"]), max(prev[1], state["i"]))
        if state["witness"]:
            TIE_WITNESSES[meta["case"]] = state["witness"]
    return m, state


GOLDEN_CASES = ["c1p_d10", "c1p_d30", "c1p_d10_weighted", "gap_d10", "sparse_d10", "voronoi_d10", "nan_d10", "grid2d_d10", "late_start_d10"]

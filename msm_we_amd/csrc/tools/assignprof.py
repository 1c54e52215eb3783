import sys, ctypes, os, numpy as np
sys.path.insert(0, '.')
from msm_we_amd import _abi
lib = ctypes.CDLL('msm_we_amd/csrc/_var/lib_assignprof' + os.environ.get('PROF_VAR', '') + '.so')
for waves in sys.argv[1:] or ['72']:
    os.environ['HAMSM_FILTER_LDS_KB'] = waves
    h = _abi.Hamsm(lib=lib)
    D, S, K, n = 30, int(os.environ.get('PROF_S', '4')), 250, int(os.environ.get("PROF_N", "2000000"))
    h.set_geometry(D, 1, S)
    b = np.arange(S + 1, dtype=np.float32); b[-1] = np.inf
    h.set_mapper_rectilinear([b]); h.set_bounds([[-np.inf, 0.05]], [[S - 0.05, np.inf]])
    rng = np.random.default_rng(0)
    for s in range(S):
        h.set_centroids(s, rng.normal(size=(K, D)))
    X = rng.normal(size=(n, D)).astype(np.float32); p = rng.uniform(0, S, size=(n, 1))
    dX, dp, dl = h.to_device(X), h.to_device(p), h.alloc(n, np.int32)
    z = np.zeros(8, dtype=np.uint64)
    h.assign(dX, 0, D, dp, n, dl); h.sync()
    out0 = np.zeros(8, dtype=np.uint64); lib.hamsm_debug_assign_profile(out0.ctypes.data_as(ctypes.c_void_p))
    import time; h.sync(); tw = time.perf_counter(); h.assign(dX, 0, D, dp, n, dl); h.sync(); tw = time.perf_counter() - tw
    out = np.zeros(8, dtype=np.uint64); lib.hamsm_debug_assign_profile(out.ctypes.data_as(ctypes.c_void_p))
    d = (out - out0).astype(float); nb = d[7]
    names = ['prologue', 'perm load + next issue', 'MFMA sweep + ranking', 'merge + store', 'commit (wait rows, LDS writes)', 'B operand']
    tot = d[:6].sum()
    if d[6] > 0: print(f'core clock while the workgroups ran: {tot / (d[6] * 10e-9) / 1e9:.2f} GHz (s_memtime cycles per 100 MHz wall tick)')
    print(f'wall {tw*1e3:.3f} ms for the call; if one workgroup per CU at a time: {tot/256/tw/1e9:.2f} G stamps/s per CU')
    print(f'filter LDS budget {waves} KB: {nb:.0f} workgroups, {tot/nb:.0f} cycles per workgroup (thread 0 view)')
    for nm, v in zip(names, d[:6]):
        print(f'  {nm:24s} {v/nb:9.0f}  {100*v/tot:5.1f}%')
    h.close()

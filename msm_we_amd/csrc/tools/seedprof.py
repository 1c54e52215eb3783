import sys, ctypes, numpy as np
sys.path.insert(0, '.')
from msm_we_amd import _abi
lib = ctypes.CDLL('msm_we_amd/csrc/_var/lib_seedprof.so')
h = _abi.Hamsm(lib=lib)
D, n, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (30, 2500, 250)
h.set_geometry(D, 1, 1)
rng = np.random.default_rng(0)
X = rng.normal(size=(n, D)).astype(np.float32)
dX = h.to_device(X)
L = 2 + int(np.log(K)); u = rng.uniform(size=(K - 1) * L)
for rep in range(3):
    h.kmeans_seed(0, dX, 0, D, None, n, None, None, K, 5, u, L)
out = np.zeros(16, dtype=np.uint64)
lib.hamsm_debug_seed_profile(out.ctypes.data_as(ctypes.c_void_p))
names = ['chunk totals', 'base scan', 'search', 'crow gather', 'distances', 'reduce+argmin', 'update']
tot = out[:7].sum()
for nm, v in zip(names, out[:7]):
    print(f'{nm:15s} {v/3/(K-1):9.0f} ticks/step  {100*v/tot:5.1f}%')
print('total ticks/step', tot/3/(K-1), '(s_memtime ticks, 100 MHz => x10 ns)')

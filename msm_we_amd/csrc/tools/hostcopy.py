#!/usr/bin/env python3
"""What the host side of staging can deliver on this box: threads copying pageable arrays into a page-locked slab (GB/s by thread
count), the upload of that slab (one copy stream), hipHostRegister of a pageable array + upload from it."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from msm_we_amd import _abi
h = _abi.Hamsm()
MB = 1 << 20
n = 512 * MB // 4
src = np.random.default_rng(0).random(n, dtype=np.float32)
pin = h.pinned(n, np.float32)
pin[:] = 0
dev = h.alloc(n, np.float32)
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in (1, 2, 4, 8, 12, 16):
    parts = np.array_split(np.arange(n), th * 4)
    def cp(ix):
        a, b = ix[0], ix[-1] + 1
        pin[a:b] = src[a:b]
    with ThreadPoolExecutor(th) as ex:
        list(ex.map(cp, parts))
        t = time.perf_counter(); list(ex.map(cp, parts)); dt = time.perf_counter() - t
    print(f"copy pageable -> pinned, {th:2d} threads: {n * 4 / dt / 1e9:6.2f} GB/s")
for rep in range(2):
    t = time.perf_counter(); h.h2d_async(dev, pin); h.copy_sync(); dt = time.perf_counter() - t
    print(f"upload 512 MB from the pinned slab: {n * 4 / dt / 1e9:6.2f} GB/s")
t = time.perf_counter(); h.h2d(dev, src); h.sync(); dt = time.perf_counter() - t
print(f"hipMemcpy from the pageable array: {n * 4 / dt / 1e9:6.2f} GB/s")
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
for sz in (64 * MB, 512 * MB):
    t = time.perf_counter(); rc = hip.hipHostRegister(src.ctypes.data, sz, 0); dt = time.perf_counter() - t
    print(f"hipHostRegister {sz // MB} MB: rc {rc}, {dt * 1e3:.1f} ms = {sz / dt / 1e9:.1f} GB/s")
    if rc == 0:
        t = time.perf_counter(); h.h2d_async(dev, src[: sz // 4]); h.copy_sync(); dt2 = time.perf_counter() - t
        print(f"   upload from the registered array: {sz / dt2 / 1e9:.2f} GB/s")
        t = time.perf_counter(); hip.hipHostUnregister(src.ctypes.data); print(f"   unregister {(time.perf_counter() - t) * 1e3:.1f} ms")

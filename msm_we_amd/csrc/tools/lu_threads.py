import numpy as np, time, os, scipy.linalg as sla
from threadpoolctl import threadpool_limits, threadpool_info
print(os.cpu_count(), len(os.sched_getaffinity(0)))
for i in threadpool_info(): print(i)
rng=np.random.default_rng(0)
n=2606
A=rng.random((n,n))
sla.lu_factor(A.copy())
for lim in (None,64,32,16,8,4,1):
    ts=[]
    for _ in range(3):
        B=A.copy(); t=time.perf_counter()
        if lim is None: sla.lu_factor(B,overwrite_a=True,check_finite=False)
        else:
            with threadpool_limits(limits=lim,user_api='blas'): sla.lu_factor(B,overwrite_a=True,check_finite=False)
        ts.append(time.perf_counter()-t)
    print(lim, min(ts), flush=True)

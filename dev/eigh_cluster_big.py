import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, time
import petal_decomposition_amd as petal
import test_gpu_eigh as te
ctx = petal.Context(0)
for d in (142, 150, 200, 260, 520):
    for kind in ("clustered", "rank_deficient"):
        for dt, tols in ((np.float64, (1e-12, 1e-11, 1e-12)), (np.float32, (4e-6, 2e-5, 3e-6))):
            t0 = time.time()
            try:
                te._check(ctx, te._data(d, kind, dt, 300 + d), tol_sigma=tols[0], tol_orth=tols[1], tol_res=tols[2])
                print(f"ok   d={d} {kind} {dt.__name__} {time.time() - t0:.1f}s", flush=True)
            except Exception as e:
                print(f"FAIL d={d} {kind} {dt.__name__}: {str(e)[:200]}", flush=True)

"""dev: the fp64 accuracy route (Cholesky-QR2 + one-sided Jacobi) at orders on both sides of its LDS / global-memory forms"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import petal_decomposition_amd as petal
import parity_cases as pc
ctx = petal.Context(0)
rng = np.random.default_rng(44)
for (n, d, k) in [(600, 10, 10), (2000, 64, 64), (2000, 96, 96), (2000, 97, 97), (3000, 128, 128), (3000, 200, 40), (4000, 300, 300), (4000, 300, 20)]:
    u, _ = np.linalg.qr(rng.standard_normal((n, d)))
    v, _ = np.linalg.qr(rng.standard_normal((d, d)))
    sig = 10.0 ** (-np.arange(d) * (6.0 / (d - 1)))          # 1 .. 1e-6
    x = (u * sig) @ v.T
    t0 = time.time()
    try:
        m = petal.PcaBuilder.new(k).centering(False).context(ctx).build().fit(x)
        se = np.abs(m.singular_values() / sig[:k] - 1.0).max()
        ce = pc.rowwise_rel(m.components(), v.T[:k]).max()
        print(("ok  " if se <= 1e-9 and ce <= 1e-8 else "FAIL"), f"n={n} d={d} k={k}: sigma rel {se:.1e} comp {ce:.1e}  {time.time() - t0:.2f}s", flush=True)
    except Exception as e:
        print(f"FAIL n={n} d={d} k={k}: {str(e)[:200]}", flush=True)

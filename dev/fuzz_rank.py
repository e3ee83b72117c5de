"""dev: rank-deficient data (rank r < l = k + 10) over the kernel ranges: the robust redo path at every Cholesky / eigen order class"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import petal_decomposition_amd as petal
import parity_cases as pc
po = pc.po
ctx = petal.Context(0)
rng = np.random.default_rng(3)
bad = 0
for (n, d, r, k, dt) in [(4096, 128, 10, 16, np.float32), (4096, 256, 40, 100, np.float32), (4096, 256, 40, 100, np.float64), (4096, 320, 60, 140, np.float32),
                         (4096, 320, 60, 140, np.float64), (4096, 320, 100, 190, np.float32), (4096, 320, 100, 190, np.float64), (3000, 400, 120, 250, np.float32),
                         (3000, 400, 120, 250, np.float64), (1000, 64, 5, 40, np.float32), (1000, 64, 5, 40, np.float64), (2000, 160, 1, 20, np.float32)]:
    x = ((rng.standard_normal((n, r)) * np.logspace(0, -1.5, r)) @ rng.standard_normal((r, d)) + rng.standard_normal(d)).astype(dt)
    om = rng.standard_normal((d, k + 10)).astype(dt)
    try:
        o = po.RandomizedPcaOracle(k, n_iter=5).fit(x.astype(np.float64), omega=om.astype(np.float64))
        m = petal.RandomizedPca(k, ctx=ctx, n_iter=5)
        y = np.asarray(m.fit_transform(x, omega=om))
        fin = np.all(np.isfinite(m.components())) and np.all(np.isfinite(y)) and np.all(np.isfinite(m.singular_values()))
        tol_s = 2e-4 if dt == np.float32 else 1e-9
        s_ok = np.allclose(m.singular_values()[:r], o.singular[:r], rtol=tol_s)
        tail_ok = np.all(m.singular_values()[r:] < (2e-3 if dt == np.float32 else 1e-7) * m.singular_values()[0])
        crel = pc.rowwise_rel(m.components()[:r].astype(np.float64), o.components[:r]).max()
        back = np.asarray(m.inverse_transform(m.transform(x)))
        b_ok = np.abs(back - x).max() < (1e-3 if dt == np.float32 else 1e-8) * np.abs(x).max()
        ok = fin and s_ok and tail_ok and crel < (2e-3 if dt == np.float32 else 1e-7) and b_ok
        print(("ok  " if ok else "FAIL"), f"n={n} d={d} rank={r} k={k} {dt.__name__}: finite {fin} sigma {s_ok} tail {tail_ok} comp {crel:.1e} roundtrip {b_ok}", flush=True)
        bad += 0 if ok else 1
    except Exception as e:
        bad += 1
        print(f"FAIL n={n} d={d} rank={r} k={k} {dt.__name__}: {str(e)[:200]}", flush=True)
print("failures:", bad)

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import petal_decomposition_amd as petal
import parity_cases as pc
po = pc.po
ctx = petal.Context(0)
def t(tag, fn):
    try: print("ok  ", tag, fn(), flush=True)
    except Exception as e: print("FAIL", tag, str(e)[:200], flush=True)
for (n, d, k, it) in [(511, 320, 221, 4), (256, 272, 227, 4), (255, 256, 212, 4), (256, 160, 144, 4), (511, 320, 187, 4)]:
    t(f"rpca f64 n={n} d={d} k={k}", lambda: pc.rpca_parity(ctx, n, d, k, it, seed=77, dtype=np.float64, tol=1e-8, device=True))
# ICA cases that stopped 45 degrees off: GPU vs oracle, iteration counts and source correlations
for (n, d, nc, seed, dt) in [(5000, 4, 3, None, np.float64), (5000, 8, 6, None, np.float32)]:
    for sd in range(4000, 4050):
        x = po.synth_ica(n, d, nc, seed=sd, dtype=dt)
        w0 = np.random.default_rng(sd + 7).standard_normal((nc, nc))
        o = po.FastIcaOracle(n_components=nc, whiten="eigh"); o.fit(x.astype(np.float64), w_init=w0)
        m = petal.FastIca(ctx=ctx, n_components=nc); m.fit(x, w_init=w0.astype(dt))
        so = np.asarray(o.transform(x.astype(np.float64))); sl = np.asarray(m.transform(x))
        c = np.abs(np.corrcoef(so.T, sl.T)[:nc, nc:]).max(axis=1).min()
        if c < 0.99: print(f"ica {dt.__name__} n={n} d={d} nc={nc} seed={sd}: n_iter oracle {o.n_iter} lib {m.n_iter} min corr {c:.3f}", flush=True)
print("done")

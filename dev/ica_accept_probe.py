"""dev: which verdict redoes a FastICA fit with orthonormal mixing directions and graded amplitudes (the accept bound of the split-product covariance)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import petal_decomposition_amd as petal
ctx = petal.Context(0)
n, d, nc = 20000, 400, 8
for gauss in (True, False):
    for amp in (0.0, 0.1, 0.25, 0.35, 0.5, 0.75):
        rng = np.random.default_rng(62 + int(100 * amp))
        s_ = rng.laplace(size=(n, nc))
        if gauss:
            a = rng.standard_normal((nc, d)) * np.logspace(0, -amp, nc)[:, None]
        else:
            q, _ = np.linalg.qr(rng.standard_normal((d, nc)))
            a = (q.T * np.sqrt(d)) * np.logspace(0, -amp, nc)[:, None]
        x = (s_ @ a + 1e-4 * rng.standard_normal((n, d))).astype(np.float32)
        lam = np.linalg.eigvalsh(np.cov(x.astype(np.float64).T))[::-1][:nc + 2]
        w0 = rng.standard_normal((nc, nc)).astype(np.float32)
        m = petal.FastIca(ctx=ctx, n_components=nc)
        m.fit(x, w_init=w0)
        st = ctx.stats()
        print(f"gauss={gauss} amp={amp}: lam ratio {lam[nc-1]/lam[0]:.3f} next {lam[nc]/lam[0]:.1e} -> redo {st['ica_redo']} split {st['ica_gram_split']} folded {st['means_folded']} iters {m.n_iter}", flush=True)

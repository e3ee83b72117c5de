"""dev: which side of a FastICA mismatch sits at the worse fixed point?  sum_j |E log cosh y_j - E log cosh g| of the unit-variance outputs"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import petal_decomposition_amd as petal
from oracle import petal_oracle as po
ctx = petal.Context(0)
n, d, nc = 20000, 128, 8
G = np.mean(np.log(np.cosh(np.random.default_rng(0).standard_normal(2_000_000))))
def contrast(y):
    y = (y - y.mean(0)) / y.std(0)
    return np.abs(np.mean(np.log(np.cosh(y)), axis=0) - G)
for seed in range(9000, 9040):
    x = po.synth_ica(n, d, nc, seed=seed, dtype=np.float64)
    w0 = np.random.default_rng(seed + 7).standard_normal((nc, nc))
    o = po.FastIcaOracle(n_components=nc, whiten="eigh"); o.fit(x, w_init=w0); yo = o.transform(x)
    m = petal.FastIca(ctx=ctx, n_components=nc); y = np.asarray(m.fit_transform(x, w_init=w0))
    c = np.abs(y.T @ yo); perm = c.argmax(axis=1)
    dev = max(np.abs(1.0 - c[np.arange(nc), perm]).max(), np.abs(c - np.eye(nc)[perm]).max()) if sorted(perm.tolist()) == list(range(nc)) else 9.0
    if dev > 5e-3:
        cl, co = contrast(y), contrast(yo)
        print(f"seed {seed}: dev {dev:.3f}; iterations lib {m.n_iter} oracle {o.n_iter}; contrast lib sum {cl.sum():.5f} min {cl.min():.5f} | oracle sum {co.sum():.5f} min {co.min():.5f}", flush=True)
        # the same fit from the oracle's point of view with the OTHER sign pattern: flip w_init's columns one at a time? (cheap check: negate w_init)
        o2 = po.FastIcaOracle(n_components=nc, whiten="eigh"); o2.fit(x, w_init=-w0); y2 = o2.transform(x)
        print(f"   oracle from -w_init: iterations {o2.n_iter}, contrast sum {contrast(y2).sum():.5f}")
print("done")

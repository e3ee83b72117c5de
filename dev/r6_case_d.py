"""dev: seed 9034 of dev/r6_case_c.py -- the oracle started from the product's side of the whitening's sign ambiguity: X1_lib = diag(s) X1_oracle
with s_j = the sign the library's convention gives eigenvector j (largest-magnitude component positive), i.e. w_init . diag(s) on the oracle's X1"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import petal_decomposition_amd as petal
from oracle import petal_oracle as po
ctx = petal.Context(0)
n, d, nc, seed = 20000, 128, 8, 9034
x = po.synth_ica(n, d, nc, seed=seed, dtype=np.float64)
w0 = np.random.default_rng(seed + 7).standard_normal((nc, nc))
o = po.FastIcaOracle(n_components=nc, whiten="eigh"); o.fit(x, w_init=w0)
k = o.k_
s = np.sign(k[np.arange(nc), np.abs(k).argmax(axis=1)])
print("sign pattern of the oracle's whitening rows under the library's convention:", s.astype(int).tolist())
o2 = po.FastIcaOracle(n_components=nc, whiten="eigh"); o2.fit(x, w_init=w0 * s[None, :]); y2 = o2.transform(x)
m = petal.FastIca(ctx=ctx, n_components=nc); y = np.asarray(m.fit_transform(x, w_init=w0))
def devn(a, b):
    c = np.abs(a.T @ b); perm = c.argmax(axis=1)
    return max(np.abs(1.0 - c[np.arange(nc), perm]).max(), np.abs(c - np.eye(nc)[perm]).max()) if sorted(perm.tolist()) == list(range(nc)) else 9.0
print(f"library vs oracle(w_init): {devn(y, o.transform(x)):.2e}; library vs oracle(w_init . diag(s)): {devn(y, y2):.2e}; iterations {m.n_iter} / {o.n_iter} / {o2.n_iter}")

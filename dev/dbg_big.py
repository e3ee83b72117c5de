import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import petal_decomposition_amd as petal
import parity_cases as pc
from oracle import petal_oracle as po
n, d, k = 20000, 1024, 256
x64 = po.synth_pca(n, d, k, seed=41, dtype=np.float64)
om = np.random.default_rng(1041).standard_normal((d, k + 10))
o = po.RandomizedPcaOracle(k, n_iter=5)
uo = o._inner_fit(x64, omega=om)
yo = po.transform_with_u(uo, o.singular, k)
ctx = petal.Context(0)
for step, dt in enumerate((np.float64, np.float32, np.float32, np.float64, np.float32)):
    x = x64.astype(dt)
    m = petal.RandomizedPca(k, ctx=ctx, n_iter=5)
    y = m.fit_transform(torch.from_numpy(x).cuda(), omega=om.astype(dt)).cpu().numpy().astype(np.float64)
    c = np.abs(np.sum((y / np.linalg.norm(y, axis=0)) * (yo / np.linalg.norm(yo, axis=0)), axis=0))
    s = np.sign(np.sum(y * yo, axis=0))
    err = np.abs(y * s - yo).max(axis=0) / np.abs(yo).max()
    print(step, dt.__name__, "min |corr|", c.min(), "n bad cols", (err > 1e-3).sum(), np.nonzero(err > 1e-3)[0][:8], "norm ratio", (np.linalg.norm(y, axis=0) / np.linalg.norm(yo, axis=0))[:4])

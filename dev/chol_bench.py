"""dev: the re-basing Cholesky alone (k_chol_inv2 through petal.chol_inv if exposed, else a few fits) -- run under dev/kt.sh"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
for (n, d, k, it, seed) in [(100000, 512, 64, 5, 2), (60000, 1024, 128, 7, 4)]:
    xd = torch.from_numpy(synth_pca(n, d, k, seed=seed, dtype=np.float32)).cuda()
    om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
    ctx = petal.Context(0)
    m = petal.RandomizedPca(k, ctx=ctx, n_iter=it)
    for rep in range(6):
        t0 = time.perf_counter(); m.fit(xd, omega=om); dt = time.perf_counter() - t0
    print(f"{n}x{d} k={k}: fit {dt*1e3:.3f} ms", flush=True)
    ctx.close()

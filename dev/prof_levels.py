"""dev: what the event brackets cost -- configs[1] fits at profiling level 0 / 1 / 2 (alternating, same process)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
n, d, k, it = 100000, 512, 64, 5
xd = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float32)).cuda()
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
ctx = petal.Context(0)
m = petal.RandomizedPca(k, ctx=ctx, n_iter=it)
for _ in range(30): m.fit(xd, omega=om)
for rnd in range(3):
    for level in (0, 1, 2):
        ctx.set_profiling(level)
        for _ in range(5): m.fit(xd, omega=om)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): m.fit(xd, omega=om)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        print(f"level {level}: {dt*1e3:.4f} ms / fit", flush=True)

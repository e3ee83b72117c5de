"""dev: fit times right after the GPU sat idle for two seconds (clock ramp)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
n, d, k = 100000, 512, 64
xd = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float32)).cuda()
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
ctx = petal.Context(0)
m = petal.RandomizedPca(k, ctx=ctx, n_iter=5)
for rnd in range(2):
    time.sleep(2.0)
    ts = []
    for i in range(120):
        t0 = time.perf_counter(); m.fit(xd, omega=om); ts.append((time.perf_counter() - t0) * 1e3)
    print("after 2 s idle:", " ".join(f"{t:.2f}" for t in ts[:24]), "... mean of fits 60-119:", f"{np.mean(ts[60:]):.4f}", "mean 10-59:", f"{np.mean(ts[10:60]):.4f}")

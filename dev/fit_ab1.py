"""whole RandomizedPca.fit at configs[1] only (alternating A/B runs: dev/ab_fast.sh)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
ctx = petal.Context(0)
n, d, k = 100000, 512, 64
x = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float32)).cuda()
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
m = petal.RandomizedPca(k, ctx=ctx, n_iter=5)
t_w = time.perf_counter()
while time.perf_counter() - t_w < 0.5: m.fit(x, omega=om)
ts = []
for rep in range(300):
    torch.cuda.synchronize(); t0 = time.perf_counter(); m.fit(x, omega=om); ts.append(time.perf_counter() - t0)
print(f"{os.path.basename(os.environ.get('PETAL_HIP_LIBRARY', 'default'))}: fit median {np.median(ts)*1e3:.4f} ms, p10 {np.percentile(ts,10)*1e3:.4f}", flush=True)

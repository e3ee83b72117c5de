"""dev: exact Pca on data WITHOUT a spectral gap behind the k wanted components (linearly decaying singular values) -- the subspace
iteration cannot converge there and the fit falls back to the full eigen-solve of the d x d covariance; for dev/timeline.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
n, d, k = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (20000, 512, 32)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(n, d, device="cuda", generator=g) * torch.linspace(3.0, 0.3, d, device="cuda")
ctx = petal.Context(0)
m = petal.Pca(k, ctx=ctx)
for rep in range(4):
    t0 = time.perf_counter(); m.fit(x); dt = time.perf_counter() - t0
    print(f"fit {dt*1e3:.3f} ms", flush=True)
    time.sleep(0.005)

"""Runs 50 FastICA iterations at the loop size of configs[2] (200000 samples, 32 components; for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
nc = 32
g = torch.Generator(device="cuda"); g.manual_seed(5)
x = torch.randn((n, nc), generator=g, device="cuda", dtype=torch.float32)
w0 = np.random.default_rng(7).standard_normal((nc, nc)).astype(np.float32)
ctx = petal.Context(0)
m = petal.FastIca(ctx=ctx, n_components=nc, tol=0.0, max_iter=50)
m.fit(x, w_init=w0)
torch.cuda.synchronize()
print("done", n, nc)

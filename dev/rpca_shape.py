"""dev: a few RandomizedPca fits of one shape (argv: n d k [n_iter]) on random data with a decaying spectrum, for dev/timeline.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
n, d, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
it = int(sys.argv[4]) if len(sys.argv) > 4 else 5
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(n, d, device="cuda", generator=g) * torch.linspace(3.0, 0.3, d, device="cuda")
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
ctx = petal.Context(0)
m = petal.RandomizedPca(k, ctx=ctx, n_iter=it)
for rep in range(6):
    t0 = time.perf_counter(); m.fit(x, omega=om); dt = time.perf_counter() - t0
    print(f"fit {dt*1e3:.3f} ms", flush=True)
    time.sleep(0.003)

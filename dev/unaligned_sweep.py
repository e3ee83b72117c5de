"""dev: device inputs whose width is not a multiple of 16 (copied into the padded layout first): fit times next to the aligned width"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
ctx = petal.Context(0)
def med(f, reps=15):
    for _ in range(5): f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
g = torch.Generator(device="cuda"); g.manual_seed(1)
for (n, d, k) in [(100000, 100, 16), (100000, 112, 16), (1000000, 100, 16), (1000000, 112, 16), (200000, 500, 64), (200000, 512, 64)]:
    x = torch.randn(n, d, device="cuda", generator=g) * torch.linspace(3.0, 0.3, d, device="cuda")
    om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
    m = petal.RandomizedPca(k, ctx=ctx, n_iter=5)
    print(f"rpca {n:8d} x {d:4d} k={k:3d}: fit {med(lambda: m.fit(x, omega=om)):.3f} ms, transform {med(lambda: m.transform(x)):.3f} ms", flush=True)
    del x

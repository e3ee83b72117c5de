"""dev: fit times over a grid of shapes -- looking for shapes whose launch heuristics leave the chip idle (a low effective rate)"""
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
print("== RandomizedPca (n_iter = 5): ms per fit, effective TB/s = 12 passes x n d 4 B / time")
for (n, d, k) in [(20000, 512, 64), (50000, 512, 64), (100000, 128, 32), (100000, 256, 64), (100000, 1024, 64), (50000, 1024, 128), (400000, 256, 32), (1000000, 64, 16), (10000, 2048, 64), (300000, 512, 16)]:
    x = torch.randn(n, d, device="cuda", generator=g) * torch.linspace(3.0, 0.3, d, device="cuda")
    om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
    m = petal.RandomizedPca(k, ctx=ctx, n_iter=5)
    t = med(lambda: m.fit(x, omega=om))
    print(f"rpca {n:8d} x {d:5d} k={k:4d}: {t:8.3f} ms   {12 * n * d * 4 / t / 1e9:6.2f} TB/s-equivalent", flush=True)
    del x
print("== exact Pca: ms per fit, Gram GFLOP/s-equivalent (n d^2 / time)")
for (n, d, k) in [(5000, 64, 8), (20000, 128, 16), (50000, 256, 32), (20000, 512, 32), (100000, 512, 64), (400000, 128, 16), (1000000, 64, 8)]:
    x = torch.randn(n, d, device="cuda", generator=g) * torch.linspace(3.0, 0.3, d, device="cuda")
    m = petal.Pca(k, ctx=ctx)
    t = med(lambda: m.fit(x))
    print(f"pca  {n:8d} x {d:5d} k={k:4d}: {t:8.3f} ms   {n * d * d / t / 1e9:6.2f} TFLOP/s-equivalent", flush=True)
    del x
print("== FastIca: ms per fit")
from synth_data import synth_ica
for (n, d, nc) in [(5000, 64, 8), (20000, 128, 16), (50000, 256, 32), (100000, 512, 64), (400000, 128, 16)]:
    x = torch.from_numpy(synth_ica(n, d, nc, seed=5, dtype=np.float32)).cuda()
    w0 = np.random.default_rng(7).standard_normal((nc, nc)).astype(np.float32)
    m = petal.FastIca(ctx=ctx, n_components=nc)
    t = med(lambda: m.fit(x, w_init=w0), 8)
    print(f"ica  {n:8d} x {d:5d} nc={nc:3d}: {t:8.3f} ms  n_iter={m.n_iter}", flush=True)
    del x

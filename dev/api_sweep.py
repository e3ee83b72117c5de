"""dev: the rest of the crate's surface -- transform / fit_transform / inverse_transform of the three models -- timed on device tensors"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca, synth_ica
ctx = petal.Context(0)
def med(f, reps=15):
    for _ in range(5): f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
for (n, d, k) in [(100000, 512, 64), (1000000, 512, 64), (200000, 256, 32)]:
    x = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float32)).cuda() if n <= 200000 else torch.randn(n, d, device="cuda")
    om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
    m = petal.RandomizedPca(k, ctx=ctx, n_iter=5); m.fit(x, omega=om)
    y = m.transform(x)
    gb = n * d * 4 / 1e9
    print(f"rpca {n}x{d} k={k}: fit {med(lambda: m.fit(x, omega=om)):.3f} ms, fit_transform {med(lambda: m.fit_transform(x, omega=om)):.3f} ms, "
          f"transform {med(lambda: m.transform(x)):.3f} ms ({gb:.2f} GB in), inverse_transform {med(lambda: m.inverse_transform(y)):.3f} ms", flush=True)
    p = petal.Pca(k, ctx=ctx); p.fit(x)
    print(f"pca  {n}x{d} k={k}: fit {med(lambda: p.fit(x)):.3f} ms, fit_transform {med(lambda: p.fit_transform(x)):.3f} ms, transform {med(lambda: p.transform(x)):.3f} ms", flush=True)
    del x, y
for (n, d, nc) in [(200000, 256, 32), (500000, 512, 64)]:
    x = torch.from_numpy(synth_ica(n, d, nc, seed=5, dtype=np.float32)).cuda()
    w0 = np.random.default_rng(7).standard_normal((nc, nc)).astype(np.float32)
    f = petal.FastIca(ctx=ctx, n_components=nc); f.fit(x, w_init=w0)
    print(f"ica  {n}x{d} nc={nc}: fit {med(lambda: f.fit(x, w_init=w0), 8):.3f} ms, fit_transform {med(lambda: f.fit_transform(x, w_init=w0), 8):.3f} ms, transform {med(lambda: f.transform(x)):.3f} ms", flush=True)

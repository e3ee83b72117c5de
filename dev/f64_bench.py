"""RandomizedPca.fit on fp64 input (100000 x 512, k = 64, n_iter = 5): the fp64-matrix-core forms of K1 / K2 (development timing)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
n, d, k = 100000, 512, 64
x = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float64)).cuda()
om = np.random.default_rng(3).standard_normal((d, k + 10))
ctx = petal.Context(0); ctx.set_profiling(2)
m = petal.RandomizedPca(k, ctx=ctx, n_iter=5)
for _ in range(3): m.fit(x, omega=om)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): m.fit(x, omega=om)
dt = (time.perf_counter() - t0) / 10
st = ctx.stats()
k1, k2 = st["xp_ms"] / st["xp_launches"], st["atb_ms"] / st["atb_launches"]
fl = 2.0 * n * d * (k + 10)
print(f"fp64 fit {dt*1e3:.2f} ms ({n/dt/1e6:.1f} M samples/s); K1 {k1*1e3:.0f} us = {fl/(k1*1e-3)/1e12:.1f} TFLOP/s fp64; K2 {k2*1e3:.0f} us = {fl/(k2*1e-3)/1e12:.1f} TFLOP/s fp64")

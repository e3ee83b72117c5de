"""BASELINE configs[3] per-GPU shard: RandomizedPca k=128 on 250000 x 1024 fp32 (development timing)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
n, d, k = 250000, 1024, 128
g = torch.Generator(device="cuda"); g.manual_seed(4)
x = torch.randn((n, d), generator=g, device="cuda")
# planted decay so the spectrum is not flat
x[:, :256] *= torch.logspace(2, 0, 256, device="cuda")
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
ctx = petal.Context(0); ctx.set_profiling(2)
m = petal.RandomizedPca(k, ctx=ctx, n_iter=7)
for _ in range(3): m.fit(x, omega=om)
t0 = time.perf_counter()
for _ in range(5): m.fit(x, omega=om)
dt = (time.perf_counter() - t0) / 5
st = ctx.stats()
print(f"fit {dt*1e3:.2f} ms  ({n/dt/1e6:.2f} M samples/s); K1 avg {st['xp_ms']/st['xp_launches']*1e3:.0f} us -> {st['pass_flops']/(st['xp_ms']/st['xp_launches']*1e-3)/1e12:.1f} TF;"
      f" K2 avg {st['atb_ms']/st['atb_launches']*1e3:.0f} us -> {st['pass_flops']/(st['atb_ms']/st['atb_launches']*1e-3)/1e12:.1f} TF")

"""FastICA at one rank's share of BASELINE configs[4]: 500000 x 512 fp32, 64 components (development timing script)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
n, d, nc = 500000, 512, 64
g = torch.Generator(device="cuda"); g.manual_seed(8)
u = torch.rand((n, nc), generator=g, device="cuda", dtype=torch.float32) - 0.5
src = -torch.sign(u) * torch.log1p(-2.0 * u.abs().clamp(max=0.4999999))
a = torch.randn((nc, d), generator=g, device="cuda", dtype=torch.float32)
x = src @ a + 0.01 * torch.randn((n, d), generator=g, device="cuda", dtype=torch.float32)
w0 = np.random.default_rng(7).standard_normal((nc, nc)).astype(np.float32)
ctx = petal.Context(0)
ctx.set_profiling(2)
m = petal.FastIca(ctx=ctx, n_components=nc)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); m.fit(x, w_init=w0); dt = time.perf_counter() - t0
    st = ctx.stats()
    step = st['ica_step_ms'] / max(st['ica_step_launches'], 1)
    print(f"fit {dt*1e3:.2f} ms  n_iter={m.n_iter}  ica_step avg {step*1e3:.1f} us -> {st['ica_step_flops']/(step*1e-3)/1e12:.1f} TFLOP/s, {st['ica_step_bytes']/(step*1e-3)/1e9:.0f} GB/s")
m2 = petal.FastIca(ctx=ctx, n_components=nc, tol=0.0, max_iter=200)
t0 = time.perf_counter(); m2.fit(x, w_init=w0); print(f"fixed 200 iterations: {(time.perf_counter()-t0)*1e3:.2f} ms")

"""FastICA at BASELINE configs[2]: 200000 x 256 fp32, n_components = 32 (development timing script)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_ica
n, d, nc = 200000, 256, 32
x = synth_ica(n, d, nc, seed=5, dtype=np.float32)
xd = torch.from_numpy(x).cuda()
w0 = np.random.default_rng(7).standard_normal((nc, nc)).astype(np.float32)
ctx = petal.Context(0)
ctx.set_profiling(2)
m = petal.FastIca(ctx=ctx, n_components=nc)
for rep in range(3):
    t0 = time.perf_counter(); m.fit(xd, w_init=w0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = ctx.stats()
    print(f"fit {dt*1e3:.2f} ms  n_iter={m.n_iter}  ica_step avg {st['ica_step_ms']/max(st['ica_step_launches'],1)*1e3:.1f} us x {st['ica_step_launches']}"
          f"  -> {st['ica_step_flops']/ (st['ica_step_ms']/max(st['ica_step_launches'],1)*1e-3)/1e12:.1f} TFLOP/s, {st['ica_step_bytes']/(st['ica_step_ms']/max(st['ica_step_launches'],1)*1e-3)/1e9:.0f} GB/s")
m2 = petal.FastIca(ctx=ctx, n_components=nc, tol=0.0, max_iter=200)   # fixed 200 iterations
t0 = time.perf_counter(); m2.fit(xd, w_init=w0); torch.cuda.synchronize(); print(f"fixed 200 iterations: {(time.perf_counter()-t0)*1e3:.2f} ms, n_iter={m2.n_iter}")

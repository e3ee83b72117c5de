"""dev: where a fit's wall time goes outside the device's busy span: Python wrapper pieces, the C side's queue / sync / epilogue stamps
(PETAL_HOST_TIMELINE), the library's own wall clock (stats.fit_ms)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PETAL_HOST_TIMELINE"] = "1"
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
n, d, k, it = 100000, 512, 64, 5
xd = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float32)).cuda()
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
ctx = petal.Context(0)
ctx.set_profiling(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
m = petal.RandomizedPca(k, ctx=ctx, n_iter=it)
for rep in range(200): m.fit(xd, omega=om)
ws, cs = [], []
for rep in range(20):
    t0 = time.perf_counter(); m.fit(xd, omega=om); dt = time.perf_counter() - t0
    ws.append(dt * 1e6); cs.append(ctx.stats()["fit_ms"] * 1e3)
print(f"python wall median {np.median(ws):.1f} us, library fit_ms median {np.median(cs):.1f} us", flush=True)
# pieces of the wrapper
def tm(f, reps=200):
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6
keep = []
print(f"describe(x) {tm(lambda: petal.describe(xd, [])):.1f} us; torch sync alone {tm(lambda: torch.cuda.current_stream(xd.device).synchronize()):.1f} us; "
      f"np.empty x4 {tm(lambda: (np.empty((k, d), np.float32), np.zeros(d, np.float32), np.empty(k, np.float32), np.zeros(1, np.float32))):.1f} us; "
      f"_host(omega) {tm(lambda: petal._host(om, petal.PETAL_F32, (d, k + 10))):.1f} us; stats() {tm(lambda: ctx.stats()):.1f} us")

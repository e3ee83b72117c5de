"""dev: where a configs[1] fit's wall time goes: C-side fit_ms (petal_stats) vs the Python mirror's call"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
n, d, k = 100000, 512, 64
xd = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float32)).cuda()
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
ctx = petal.Context(0, stream=torch.cuda.current_stream().cuda_stream)
m = petal.RandomizedPca(k, ctx=ctx, n_iter=5)
for _ in range(10): m.fit(xd, omega=om)
tc = tp = 0.0
N = 50
for _ in range(N):
    t0 = time.perf_counter(); m.fit(xd, omega=om); tp += time.perf_counter() - t0
    tc += ctx.stats()["fit_ms"]
print(f"python wall {tp/N*1e3:.4f} ms  C-side fit_ms {tc/N:.4f} ms  -> mirror overhead {(tp/N*1e3 - tc/N)*1e3:.1f} us")

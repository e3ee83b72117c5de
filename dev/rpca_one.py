"""dev: a few RandomizedPca fits at configs[1] (or the configs[3] share with argv 'cfg4') separated by idle gaps, for dev/timeline.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
big = len(sys.argv) > 1 and sys.argv[1] == "cfg4"
n, d, k, it = (250000, 1024, 128, 7) if big else (100000, 512, 64, 5)
xd = torch.from_numpy(synth_pca(n, d, k, seed=4 if big else 2, dtype=np.float32)).cuda()
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
ctx = petal.Context(0)
if len(sys.argv) > 2: ctx.set_gemm_mode(sys.argv[2])
m = petal.RandomizedPca(k, ctx=ctx, n_iter=it)
for rep in range(8):
    t0 = time.perf_counter(); m.fit(xd, omega=om); dt = time.perf_counter() - t0
    print(f"fit {dt*1e3:.3f} ms", flush=True)
    time.sleep(0.002)

"""dev: a few exact Pca fits at configs[0] (1000 x 16 f64) or a tall case (argv 'tall': 200000 x 256 f32, k = 32), for dev/timeline.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
tall = len(sys.argv) > 1 and sys.argv[1] == "tall"
n, d, k, dt = (200000, 256, 32, np.float32) if tall else (1000, 16, 4, np.float64)
xd = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=dt)).cuda()
ctx = petal.Context(0)
m = petal.Pca.new(k, ctx)
for rep in range(8):
    t0 = time.perf_counter(); m.fit(xd); dt_ = time.perf_counter() - t0
    print(f"fit {dt_*1e3:.3f} ms", flush=True)
    time.sleep(0.002)

"""dev: a few FastICA fits at configs[2] (or cfg5 share with argv 'cfg5') separated by idle gaps, for dev/timeline.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_ica
big = len(sys.argv) > 1 and sys.argv[1] == "cfg5"
n, d, nc = (500000, 512, 64) if big else (200000, 256, 32)
if len(sys.argv) > 1 and sys.argv[1] == "small": n, d, nc = 20000, 256, 32
xd = torch.from_numpy(synth_ica(n, d, nc, seed=8 if big else 5, dtype=np.float32)).cuda()
w0 = np.random.default_rng(7).standard_normal((nc, nc)).astype(np.float32)
ctx = petal.Context(0)
m = petal.FastIca(ctx=ctx, n_components=nc)
for rep in range(6):
    t0 = time.perf_counter(); m.fit(xd, w_init=w0); dt = time.perf_counter() - t0
    print(f"fit {dt*1e3:.3f} ms n_iter={m.n_iter}", flush=True)
    time.sleep(0.002)

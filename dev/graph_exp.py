"""dev: stream launches against a captured graph of the same fit (PETAL_GRAPH_EXPERIMENT: 1 = capture + replay, 2 = events around
the stream-launched region); device time of the region either way"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
n, d, k, it = 100000, 512, 64, 5
xd = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float32)).cuda()
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
ctx = petal.Context(0)
ctx.set_profiling(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
m = petal.RandomizedPca(k, ctx=ctx, n_iter=it)
for rep in range(30): m.fit(xd, omega=om)
ref = np.array(m.components_) if hasattr(m, "components_") else None
for rnd in range(3):
    for mode in ("2", "1"):
        os.environ["PETAL_GRAPH_EXPERIMENT"] = mode
        for rep in range(3):
            t0 = time.perf_counter(); m.fit(xd, omega=om); dt = time.perf_counter() - t0
            print(f"mode {mode}: fit {dt*1e3:.3f} ms", flush=True)
os.environ.pop("PETAL_GRAPH_EXPERIMENT")

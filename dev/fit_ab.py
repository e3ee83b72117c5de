"""RandomizedPca.fit wall time (device-resident X, warm): configs[1] and the north-star point; run under PETAL_NO_POW3=1 for the A/B"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import petal_decomposition_amd as petal
from synth_data import synth_pca_device

ctx = petal.Context(0)
for (n, d, k, n_iter, reps) in ((100000, 512, 64, 5, 60), (1000000, 512, 64, 5, 12), (1000000, 512, 64, 7, 8)):
    x = synth_pca_device(n, d, k, 3, 0, n, "cuda")
    om = np.random.default_rng(5).standard_normal((d, k + 10)).astype(np.float32)
    m = petal.RandomizedPca(k, ctx=ctx, n_iter=n_iter)
    for _ in range(5):
        m.fit(x, omega=om)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.fit(x, omega=om)
        ts.append(time.perf_counter() - t0)
    st = ctx.stats()
    print(f"{n}x{d} k={k} n_iter={n_iter}: fit median {np.median(ts)*1e3:.3f} ms, min {np.min(ts)*1e3:.3f} ms; redo={st['rpca_redo']} knobs={ {k2: v for k2, v in os.environ.items() if k2.startswith('PETAL_')} }", flush=True)
    del x
ctx.close()

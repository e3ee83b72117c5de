"""fused power-iteration pass (k_pow3) against K1 + K2: kernel times from the ctx's event brackets (profiling level 2)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import petal_decomposition_amd as petal

def run(n, N, mode_env):
    ctx = petal.Context(0)
    ctx.set_profiling(2)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randn((n, 512), generator=g, device="cuda", dtype=torch.float32)
    p = np.random.default_rng(2).standard_normal((512, N)).astype(np.float32)
    mu = np.random.default_rng(3).standard_normal(512).astype(np.float32)
    res = []
    for rep in range(6):
        y, z, fused = petal.power_pass(x, p, mu, want_z=False, ctx=ctx)
        st = ctx.stats()
        res.append((fused, st["pow_ms"], st["pow_launches"], st["xp_ms"], st["atb_ms"]))
    print(f"n={n} N={N}: fused={res[-1][0]} pow_ms={np.median([r[1] for r in res[2:]])*1e3:.1f} us, K1 {np.median([r[3] for r in res[2:]])*1e3:.1f} us, K2 {np.median([r[4] for r in res[2:]])*1e3:.1f} us", flush=True)
    ctx.close()

for n in (100000, 1000000):
    for N in (74, 64):
        run(n, N, None)

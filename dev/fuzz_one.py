import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import petal_decomposition_amd as petal
import parity_cases as pc
po = pc.po
ctx = petal.Context(0, lib=petal.load_library(os.environ["PETAL_LIB"])) if os.environ.get("PETAL_LIB") else petal.Context(0)
if len(sys.argv) > 1 and sys.argv[1] in ("fp32", "bf16x3"): ctx.set_gemm_mode(sys.argv[1])
cases = [(4096, 160, 140, 4), (4096, 320, 140, 4), (4096, 320, 131, 4), (4096, 320, 130, 4), (4096, 160, 100, 4), (4096, 320, 180, 4), (4096, 320, 140, 1), (4096, 320, 140, 2)]
for (n, d, k, it) in cases:
    seed = 1079
    for dt in (np.float32, np.float64):
        x = po.synth_pca(n, d, k, seed=seed, dtype=dt)
        om = np.random.default_rng(seed + 1000).standard_normal((d, k + 10))
        o = po.RandomizedPcaOracle(k, n_iter=it); o._inner_fit(x.astype(np.float64), omega=om)
        m = petal.RandomizedPca(k, ctx=ctx, n_iter=it); m.fit(torch.from_numpy(x).cuda(), omega=om.astype(dt))
        rel = pc.rowwise_rel(m.components().astype(np.float64), o.components)
        srel = np.abs(m.singular_values() / o.singular - 1)
        print(f"{dt.__name__} n={n} d={d} k={k} it={it}: comp rel max {rel.max():.2e} (idx {rel.argmax()}), sigma rel max {srel.max():.2e} (idx {srel.argmax()})", flush=True)

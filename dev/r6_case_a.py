"""dev: one uncentred off-centre case of dev/fuzz_round6.py under the precision options"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import petal_decomposition_amd as petal
import parity_cases as pc
from oracle import petal_oracle as po
n, d, k = 33333, 1024, 100
off, cent = 3.0, False
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 9500
x = po.synth_pca(n, d, k, seed=seed, dtype=np.float64)
x = (x + off * x.std(axis=0) * np.sign(np.random.default_rng(7).standard_normal(d))).astype(np.float32)
om = np.random.default_rng(seed + 1000).standard_normal((d, k + 10)).astype(np.float32)
for it in (3, 4, 5, 7):
    o = po.RandomizedPcaOracle(k, centering=cent, n_iter=it); o._inner_fit(x.astype(np.float64), omega=om.astype(np.float64))
    o32 = po.RandomizedPcaOracle(k, centering=cent, n_iter=it); o32._inner_fit(x, omega=om)
    e32 = pc.rowwise_rel(o32.components.astype(np.float64), o.components)
    line = f"n_iter={it}: fp32 oracle {e32.max():.2e}"
    for name, opts, mode in (("default", {}, None), ("no steering", {"steering_passes": 0}, None), ("exact planes", {}, "bf16x3-exact"), ("fp32 mfma", {}, "fp32")):
        ctx = petal.Context(0)
        if mode: ctx.set_gemm_mode(mode)
        for kk, v in opts.items(): ctx.set_option(kk, v)
        m = petal.RandomizedPca(k, centering=cent, ctx=ctx, n_iter=it).fit(x, omega=om)
        rel = pc.rowwise_rel(m.components().astype(np.float64), o.components)
        line += f" | {name}: {rel.max():.2e} (worst row {int(rel.argmax())}, redo {ctx.stats()['rpca_redo']} eigh_redo {ctx.stats()['eigh_redo']})"
        ctx.close()
    print(line, flush=True)

"""n_iter 0..3 parity against the oracle (same Omega), both GEMM modes; prints the component / sigma errors.  Run under PETAL_NO_P2_OMEGA=1
or PETAL_NO_P2_ITERATE=1 to see which of the two-plane operands an error comes from."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import petal_decomposition_amd as petal
import parity_cases as pc
from oracle import petal_oracle as po

print("knobs:", {k: v for k, v in os.environ.items() if k.startswith("PETAL_")})
for mode in ("bf16x3", "fp32"):
    ctx = petal.Context(0)
    ctx.set_gemm_mode(mode)
    for (n, d, k) in ((4000, 256, 16), (20000, 512, 64)):
        for spectrum in ("planted", "geo97", "rsqrt"):
            for n_iter in (0, 1, 2, 3):
                seed = 300 + n_iter
                x = po.synth_pca(n, d, k, seed=seed, dtype=np.float32) if spectrum == "planted" else pc.slow_decay_matrix(n, d, spectrum, seed, np.float32)
                om = np.random.default_rng(seed + 1000).standard_normal((d, k + 10)).astype(np.float32)
                o = po.RandomizedPcaOracle(k, n_iter=n_iter).fit(x.astype(np.float64), omega=om.astype(np.float64))
                m = petal.RandomizedPca(k, ctx=ctx, n_iter=n_iter).fit(x, omega=om)
                rel = pc.rowwise_rel(m.components().astype(np.float64), o.components)
                sg = np.abs(m.singular_values() / o.singular - 1)
                print(f"{mode:7s} {n}x{d} k={k} {spectrum:8s} n_iter={n_iter}: comp max {rel.max():.2e} (row {rel.argmax()}), lead-half {rel[:k//2].max():.2e}, sigma {sg.max():.2e}", flush=True)
    ctx.close()

"""component errors against the oracle with and without the steering products (PETAL_NO_POW3_FAST), a few shapes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import petal_decomposition_amd as petal
from oracle import petal_oracle as po
import parity_cases as pc
ctx = petal.Context(0)
SHAPES = ((20000, 512, 64, 5, 91), (33333, 500, 24, 3, 92), (8192, 512, 64, 7, 93), (100000, 512, 64, 5, 7), (20000, 512, 64, 3, 5)) if len(sys.argv) > 1 else ((20000, 1024, 128, 4, 96), (9000, 400, 100, 7, 97), (20000, 1024, 128, 7, 99), (40000, 1024, 64, 5, 100), (9000, 400, 40, 5, 101))
for (n, d, k, it, seed) in SHAPES:
    x = po.synth_pca(n, d, k, seed=seed, dtype=np.float32)
    om = np.random.default_rng(seed + 1000).standard_normal((d, k + 10)).astype(np.float32)
    o = po.RandomizedPcaOracle(k, n_iter=it)
    o._inner_fit(x.astype(np.float64), omega=om.astype(np.float64))
    out = []
    for knob in (None, "1"):
        if knob: os.environ["PETAL_NO_POW3_FAST"] = knob
        else: os.environ.pop("PETAL_NO_POW3_FAST", None)
        m = petal.RandomizedPca(k, ctx=ctx, n_iter=it).fit(x, omega=om)
        rel = pc.rowwise_rel(m.components().astype(np.float64), o.components)
        out.append((rel.max(), int(rel.argmax()), np.median(rel), ctx.stats()["rpca_redo"]))
    os.environ.pop("PETAL_NO_POW3_FAST", None)
    print(f"{n}x{d} k={k} n_iter={it}: steering: max {out[0][0]:.2e} (row {out[0][1]}, median {out[0][2]:.1e}, redo {out[0][3]}) | exact passes: max {out[1][0]:.2e} (row {out[1][1]}, median {out[1][2]:.1e}, redo {out[1][3]})", flush=True)

"""dev: exact Pca, wide uncentred off-centre fp32 case of dev/fuzz_round6.py (seed 81): find the case index, then print the singular values"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import petal_decomposition_amd as petal
import parity_cases as pc
from oracle import petal_oracle as po
ctx = petal.Context(0)
n, d, k = 50, 256, 22
for off in (0.0, 3.0, 40.0):
    for cent in (True, False):
        for dt in (np.float32, np.float64):
            x = po.synth_pca(n, d, k, seed=9711, dtype=np.float64)
            x = (x + off * x.std(axis=0) * np.sign(np.random.default_rng(11).standard_normal(d))).astype(dt)
            o = po.PcaOracle(k, centering=cent, thin=True); o._inner_fit(x.astype(np.float64))
            m = petal.Pca(k, centering=cent, ctx=ctx); m.fit(x)
            rel = pc.rowwise_rel(m.components().astype(np.float64), o.components)
            s = np.asarray(m.singular_values(), dtype=np.float64)
            print(f"off={off} cent={cent} {dt.__name__}: comp err max {rel.max():.2e} (rows >1e-3: {np.nonzero(rel > 1e-3)[0].tolist()}), sigma lib {s[:3]} .. {s[-2:]}, oracle {o.singular[:3]} .. {o.singular[-2:]}", flush=True)

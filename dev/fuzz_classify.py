"""dev: the failures of `dev/fuzz_rpca.py <seed> <cases>` classified: for every case that misses its
tolerance against the fp64 oracle, the ORACLE itself is run on the float32 input (RandomizedPcaOracle in the data's type: the
reference's own arithmetic, LAPACK's s-routines) with the same Omega.  A miss is "data conditioning" only if that fp32 oracle misses
the fp64 one by at least half of the product's error; every other miss is the product's and becomes a regression test.
usage: python dev/fuzz_classify.py [seed=13] [cases=60]  ->  table on stdout"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import petal_decomposition_amd as petal
import parity_cases as pc
from oracle import petal_oracle as po
ctx = petal.Context(0)
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 13
rng = np.random.default_rng(seed0)
rows = []
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    d = int(rng.choice([16, 24, 48, 64, 100, 128, 160, 200, 256, 272, 320]))
    n = int(rng.choice([70, 255, 256, 257, 511, 1000, 3001, 4096, 7777]))
    kmax = max(1, min(n, d) - 10)
    k = int(rng.integers(1, min(kmax, 140) + 1))
    n_iter = int(rng.choice([1, 2, 4, 7]))
    device = bool(rng.integers(0, 2))
    cent = bool(rng.integers(0, 4) > 0)
    gap = 1.0 - 10.0 ** (-3.0 / max(k, 1))
    tol = max(2e-5, 3e-6 / gap)
    seed = 1000 + case
    x = po.synth_pca(n, d, k, seed=seed, dtype=np.float32)
    om = np.random.default_rng(seed + 1000).standard_normal((d, k + 10)).astype(np.float32)
    o = po.RandomizedPcaOracle(k, centering=cent, n_iter=n_iter)
    o._inner_fit(x.astype(np.float64), omega=om.astype(np.float64))
    xin = torch.from_numpy(x).cuda() if device else x
    m = petal.RandomizedPca(k, centering=cent, ctx=ctx, n_iter=n_iter).fit(xin, omega=om)
    err = pc.rowwise_rel(m.components().astype(np.float64), o.components).max()
    serr = np.abs(m.singular_values() / o.singular - 1).max()
    if err <= tol and serr <= 5e-5:
        continue
    o32 = po.RandomizedPcaOracle(k, centering=cent, n_iter=n_iter)
    o32._inner_fit(x, omega=om)
    e32 = pc.rowwise_rel(o32.components.astype(np.float64), o.components).max()
    s32 = np.abs(o32.singular.astype(np.float64) / o.singular - 1).max()
    cond = e32 >= 0.5 * err and (serr <= 5e-5 or s32 >= 0.5 * serr)
    rows.append(f"case {case:2d} n={n} d={d} k={k} it={n_iter} dev={int(device)} cent={int(cent)} l/min(n,d)={(k + 10) / min(n, d):.2f}: product {err:.2e} (sigma {serr:.1e}), "
                f"fp32 oracle {e32:.2e} (sigma {s32:.1e}), tol {tol:.1e} -> {'data conditioning' if cond else 'PRODUCT'}")
print("\n".join(rows))
print(f"{len(rows)} misses of the flat tolerance, {sum('PRODUCT' in r for r in rows)} of them the product's")

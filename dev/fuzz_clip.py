"""dev: RandomizedPca where l = k + 10 is clipped by n or d (pca.rs:710/713 slices), both dtypes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import petal_decomposition_amd as petal
import parity_cases as pc
ctx = petal.Context(0)
rng = np.random.default_rng(5)
bad = 0
for case in range(60):
    n = int(rng.choice([12, 30, 64, 70, 100, 300, 1000])); d = int(rng.choice([8, 16, 20, 48, 64, 100, 160]))
    m = min(n, d)
    k = int(rng.integers(max(1, m - 9), m + 1))
    if k >= n: k = n - 1   # (centring leaves rank n - 1: the n-th direction has sigma = 0 and an arbitrary vector, in the oracle too)
    dt = np.float64 if rng.integers(0, 2) else np.float32
    dev = bool(rng.integers(0, 2))
    try:
        # all of min(n, d) directions are in the block: the result is the exact SVD; the last components sit at the noise floor
        tol = 1e-7 if dt == np.float64 else 2e-3
        r = pc.rpca_parity(ctx, n, d, k, 4, seed=6000 + case, dtype=dt, tol=tol, tol_sigma=1e-8 if dt == np.float64 else 5e-5, device=dev)
        print(f"ok   n={n} d={d} k={k} {dt.__name__} dev={dev} rel={r:.1e}", flush=True)
    except Exception as e:
        bad += 1
        print(f"FAIL n={n} d={d} k={k} {dt.__name__} dev={dev}: {str(e)[:160]}", flush=True)
print("failures:", bad)

"""dev: RandomizedPca against the oracle on random shapes (same Omega): d a multiple of 16 or not, k from 1 to min(n, d) - 10,
row counts around the 256-row workgroups, host and device inputs"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import petal_decomposition_amd as petal
import parity_cases as pc
ctx = petal.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    d = int(rng.choice([16, 24, 48, 64, 100, 128, 160, 200, 256, 272, 320]))
    n = int(rng.choice([70, 255, 256, 257, 511, 1000, 3001, 4096, 7777]))
    kmax = max(1, min(n, d) - 10)
    k = int(rng.integers(1, min(kmax, 140) + 1))
    n_iter = int(rng.choice([1, 2, 4, 7]))
    device = bool(rng.integers(0, 2))
    cent = bool(rng.integers(0, 4) > 0)
    try:
        # crowded planted spectra (k close to d) pin the last vectors loosely: the tolerance follows the neighbour gap 10^(-3/k)
        gap = 1.0 - 10.0 ** (-3.0 / max(k, 1))
        tol = max(2e-5, 3e-6 / gap)
        rel = pc.rpca_parity(ctx, n, d, k, n_iter, seed=1000 + case, tol=tol, tol_sigma=5e-5, device=device, centering=cent)
        print(f"ok   n={n} d={d} k={k} it={n_iter} dev={device} cent={cent} rel={rel:.2e} tol={tol:.1e}", flush=True)
    except Exception as e:
        bad += 1
        print(f"FAIL n={n} d={d} k={k} it={n_iter} dev={device} cent={cent}: {str(e)[:200]}", flush=True)
print("failures:", bad)

"""dev: random-shape sweeps of the three fits against the oracle (converged settings, tolerances that follow the data type and
the planted spectrum's gaps).  usage: python dev/fuzz_all.py <seed> <cases> [rpca|pca|ica|icapar]"""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import petal_decomposition_amd as petal
import parity_cases as pc
ctx = petal.Context(0)
if os.environ.get("FUZZ_GEMM"): ctx.set_gemm_mode(os.environ["FUZZ_GEMM"])
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
which = sys.argv[3] if len(sys.argv) > 3 else "all"
rng = np.random.default_rng(seed0)
bad = 0
def run(tag, fn):
    global bad
    try:
        r = fn()
        print(f"ok   {tag} -> {r}", flush=True)
    except Exception as e:
        bad += 1
        print(f"FAIL {tag}: {str(e)[:160]}", flush=True)
for case in range(ncase):
    dt = np.float32 if rng.integers(0, 3) else np.float64
    if which in ("all", "rpca"):
        d = int(rng.choice([16, 24, 48, 64, 100, 128, 160, 200, 256, 272, 320, 400]))
        n = int(rng.choice([255, 256, 257, 511, 1000, 3001, 4096, 7777]))
        k = int(rng.integers(1, max(2, min(min(n, d) - 10, 230))))
        it = int(rng.choice([4, 7]))
        dev = bool(rng.integers(0, 2)); cent = bool(rng.integers(0, 4) > 0)
        gap = 1.0 - 10.0 ** (-3.0 / max(k, 1))
        tol = 1e-8 if dt == np.float64 else max(3e-5, 3e-5 / gap)
        run(f"rpca {dt.__name__} n={n} d={d} k={k} it={it} dev={dev} cent={cent} tol={tol:.0e}",
            lambda: f"{pc.rpca_parity(ctx, n, d, k, it, seed=2000 + case, dtype=dt, tol=tol, tol_sigma=1e-8 if dt == np.float64 else 3e-5, device=dev, centering=cent):.2e}")
    if which in ("all", "pca"):
        d = int(rng.choice([3, 16, 24, 64, 89, 100, 128, 200, 256, 300]))
        n = int(rng.choice([50, 255, 1000, 3001, 8000]))
        k = int(rng.integers(1, max(2, min(n, d, 64))))
        tol = 1e-8 if dt == np.float64 else 5e-5
        run(f"pca {dt.__name__} n={n} d={d} k={k}", lambda: f"{pc.pca_parity(ctx, n, d, k, seed=3000 + case, dtype=dt, tol=tol, thin_oracle=True)}")
    if which in ("all", "ica"):
        d = int(rng.choice([4, 8, 16, 32, 64, 100, 128, 256]))
        n = int(rng.choice([2000, 5000, 20000, 50001]))
        nc = int(rng.integers(2, min(d, 48) + 1))
        dev = bool(rng.integers(0, 2))
        run(f"ica {dt.__name__} n={n} d={d} nc={nc} dev={dev}",
            lambda: f"{pc.ica_parity(ctx, n, d, nc, seed=4000 + case, dtype=dt, n_components=nc, device=dev)}")
    if which in ("all", "icapar"):
        nc = int(rng.choice([2, 3, 5, 8, 16, 24, 32, 40, 64]))
        n = int(rng.choice([3000, 20000, 50001]))
        run(f"ica_par {dt.__name__} n={n} nc={nc}", lambda: f"{pc.ica_par_parity(ctx, n, nc, seed=5000 + case, dtype=dt, tol=1e-4 if dt == np.float32 else 1e-7)}")
print("failures:", bad)

"""split-product Gram in the form of PETAL_GRAM_FORM (5: k_gram5, 4: k_gram4, 3: k_gram3) through the PETAL_GRAM_SPLIT=1 hook: accuracy against
float64, wall time of the call (incl. the d2h of C); kernel times: run under dev/kt.sh"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import petal_decomposition_amd as petal
from synth_data import synth_ica
os.environ["PETAL_GRAM_SPLIT"] = "1"
ctx = petal.Context(0)
shapes = ((200000, 256, 32, 5), (500000, 512, 64, 8), (100001, 384, 32, 2), (60000, 1000, 32, 3), (20011, 200, 16, 4))
if len(sys.argv) > 1 and sys.argv[1] == "short": shapes = shapes[:2]
for (n, d, nc, seed) in shapes:
    x = synth_ica(n, d, nc, seed=seed, dtype=np.float32)
    mu = x.astype(np.float64).mean(0)
    xc = x.astype(np.float64) - mu.astype(np.float32).astype(np.float64)
    cref = xc.T @ xc
    lam = np.linalg.eigvalsh(cref)[::-1]
    xd = torch.from_numpy(x).cuda()
    mu32 = mu.astype(np.float32)
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c = petal.gemm_atb(xd, None, mu32, mu32, ctx=ctx)
        ts.append(time.perf_counter() - t0)
    err = np.abs(c - cref).max() / np.abs(cref).max()
    sym = np.abs(c - c.T).max()
    lam2 = np.linalg.eigvalsh(c)[::-1]
    c0 = petal.gemm_atb(xd, None, None, None, ctx=ctx)
    x64 = x.astype(np.float64)
    err0 = np.abs(c0 - x64.T @ x64).max() / np.abs(x64.T @ x64).max()
    print(f"   uncentred: max|dC|/max|C| {err0:.2e}")
    print(f"form={os.environ.get('PETAL_GRAM_FORM','5')} {n}x{d}: call {np.median(ts[2:])*1e3:.3f} ms; max|dC|/max|C| {err:.2e}; asym {sym:.1e}; rel err top-nc eigenvalues {np.abs(lam2[:nc]/lam[:nc]-1).max():.2e}", flush=True)

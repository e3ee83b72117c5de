"""split-product Gram (k_gram3, PETAL_GRAM_SPLIT=1 hook) against float64 and against the fp64-MFMA Gram: time and error"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import petal_decomposition_amd as petal
from synth_data import synth_ica
ctx = petal.Context(0)
for (n, d, nc, seed) in ((200000, 256, 32, 5), (500000, 512, 64, 8)):
    x = synth_ica(n, d, nc, seed=seed, dtype=np.float32)
    mu = x.astype(np.float64).mean(0)
    xc = x.astype(np.float64) - mu.astype(np.float32).astype(np.float64)
    cref = xc.T @ xc
    lam = np.linalg.eigvalsh(cref)[::-1]
    xd = torch.from_numpy(x).cuda()
    mu32 = mu.astype(np.float32)
    for hook in ("1", None):
        if hook: os.environ["PETAL_GRAM_SPLIT"] = hook
        else: os.environ.pop("PETAL_GRAM_SPLIT", None)
        ts = []
        for rep in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c = petal.gemm_atb(xd, None, mu32, mu32, ctx=ctx)
            ts.append(time.perf_counter() - t0)
        err = np.abs(c - cref).max() / np.abs(cref).max()
        lam2 = np.linalg.eigvalsh(c)[::-1]
        print(f"{n}x{d} hook={hook}: call {np.median(ts[2:])*1e3:.3f} ms (host wall incl. d2h of C); max|dC|/max|C| {err:.2e}; rel err top-nc eigenvalues {np.abs(lam2[:nc]/lam[:nc]-1).max():.2e}", flush=True)

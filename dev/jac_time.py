import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
lib = petal.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), sys.argv[1])) if len(sys.argv) > 1 else None
ctx = petal.Context(0, lib=lib) if lib else petal.Context(0)
for d in (74, 138):
    x = torch.randn((4000, d), device="cuda", dtype=torch.float64) * torch.logspace(0, -3, d, device="cuda", dtype=torch.float64)
    m = petal.Pca(d, ctx=ctx)
    m.fit(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m.fit(x)
    torch.cuda.synchronize(); print(d, "Pca.fit (full Jacobi of d x d) ms:", (time.perf_counter() - t0) / 5 * 1e3)

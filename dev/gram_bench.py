"""Exact Pca.fit on n x 512 fp32 (development timing of the fp64 Gram kernel; run under rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
d = 512
g = torch.Generator(device="cuda"); g.manual_seed(3)
x = torch.randn((n, 64), generator=g, device="cuda") @ torch.randn((64, d), generator=g, device="cuda") + 0.1 * torch.randn((n, d), generator=g, device="cuda")
ctx = petal.Context(0)
m = petal.Pca(64, ctx=ctx)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); m.fit(x); dt = time.perf_counter() - t0
    print(f"Pca.fit {n}x{d}: {dt*1e3:.2f} ms")

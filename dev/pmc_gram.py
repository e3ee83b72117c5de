"""Runs the fp64-MFMA Gram matrix of the FastICA whitening a few times at one rank's share of configs[4] (500000 x 512 fp32; for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
d = 512
g = torch.Generator(device="cuda"); g.manual_seed(8)
x = torch.randn((n, d), generator=g, device="cuda", dtype=torch.float32)
ctx = petal.Context(0)
m = petal.Pca(4, ctx=ctx)     # exact Pca: precise (fp64) Gram of the fp32 data + the subspace iteration
for _ in range(3):
    m.fit(x)
torch.cuda.synchronize()
print("done", n, d)

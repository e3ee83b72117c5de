"""Runs the two power-iteration GEMM kernels a few times at a given shape (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d, l = 512, 74
ctx = petal.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn((n, d), generator=g, device="cuda")
z = torch.randn((n, 80), generator=g, device="cuda"); z[:, l:] = 0
p = np.random.default_rng(7).standard_normal((d, l)).astype(np.float32)
mu = np.random.default_rng(8).standard_normal(d).astype(np.float32)
for _ in range(3):
    petal.gemm_xp(x, p, mu, ctx=ctx)
    petal.gemm_atb(x, z, mu, ctx=ctx)
    petal.power_pass(x, p, mu, ctx=ctx)      # the fused pass (k_pow3) in the split-product mode, K1 + K2 otherwise
torch.cuda.synchronize()
print("done", n)

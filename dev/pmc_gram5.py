"""A few split-product Gram launches (k_gram5 through the PETAL_GRAM_SPLIT=1 hook, means given) for rocprofv3 --pmc passes:
usage: python dev/pmc_gram5.py [rows] [features]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
os.environ["PETAL_GRAM_SPLIT"] = "1"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = torch.Generator(device="cuda"); g.manual_seed(8)
x = torch.randn((n, d), generator=g, device="cuda", dtype=torch.float32)
mu = np.zeros(d, dtype=np.float32)
ctx = petal.Context(0)
for rep in range(4):
    c = petal.gemm_atb(x, None, mu, mu, ctx=ctx)
torch.cuda.synchronize()
print("done", n, d)

"""accuracy of the fp64 Gram route pieces (development)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import petal_decomposition_amd as petal
rng = np.random.default_rng(44)
n, d = 3000, 16
u, _ = np.linalg.qr(rng.standard_normal((n, d)))
v, _ = np.linalg.qr(rng.standard_normal((d, d)))
sig = 10.0 ** (-np.arange(d) / 2.0)
x = (u * sig) @ v.T
ctx = petal.Context(0)
g = petal.gemm_atb(x, ctx=ctx)
gref = (x.astype(np.longdouble).T @ x.astype(np.longdouble)).astype(np.float64)
print("gemm_atb f64 max abs err / max|G|:", np.abs(g - gref).max() / np.abs(gref).max())
lam = np.linalg.eigvalsh(g)[::-1]
print("numpy eigvalsh of device G: rel err of sqrt(lam):", np.abs(np.sqrt(np.maximum(lam, 0)) / sig - 1))
lam = np.linalg.eigvalsh(gref)[::-1]
print("numpy eigvalsh of exact G: rel err of sqrt(lam):", np.abs(np.sqrt(np.maximum(lam, 0)) / sig - 1))

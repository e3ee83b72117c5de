"""Times K1 (gemm_xp) and K2 (gemm_atb) alone at a given row count and checks them against a float64 product.
usage: [PETAL_GEMM=bf16x3|fp32] python dev/k12_bench.py [rows] [d] [l]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
l = int(sys.argv[3]) if len(sys.argv) > 3 else 74
ctx = petal.Context(0, lib=petal.load_library(os.environ["PETAL_LIB"])) if os.environ.get("PETAL_LIB") else petal.Context(0)
ctx.set_profiling(2)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn((n, d), generator=g, device="cuda") * 2 + 0.5
p = np.random.default_rng(7).standard_normal((d, l)).astype(np.float32)
mu = x[:4096].mean(0).cpu().numpy().astype(np.float32)
for name in ("K1", "K2"):
    ms, cnt = 0.0, 0
    for it in range(8):
        if name == "K1":
            z = petal.gemm_xp(x, p, mu, ctx=ctx)
            st = ctx.stats(); key = "xp"
        else:
            y = petal.gemm_atb(x, z, mu, ctx=ctx)
            st = ctx.stats(); key = "atb"
        if it >= 3:
            ms += st[key + "_ms"]; cnt += st[key + "_launches"]
    avg = ms / max(cnt, 1)
    fl = 2.0 * n * d * l
    print(f"{name}: {avg*1e3:8.1f} us  {fl/(avg*1e-3)/1e12:7.1f} TFLOP/s-equivalent  {4.0*(n*d+n*l+d*l)/(avg*1e-3)/1e9:7.0f} GB/s")
# accuracy on a slice vs float64
xs = x[:2048].cpu().numpy().astype(np.float64); zs = (z[:2048].cpu().numpy() if torch.is_tensor(z) else np.asarray(z)[:2048])
ref = (xs - mu.astype(np.float64)) @ p.astype(np.float64)
print("K1 max err / mean|z| =", np.abs(zs - ref).max() / np.abs(ref).mean())
zf = (z.cpu().numpy() if torch.is_tensor(z) else np.asarray(z)).astype(np.float64)
if n <= 200000:
    yref = (x.cpu().numpy().astype(np.float64) - mu).T @ zf
    yy = y.cpu().numpy() if torch.is_tensor(y) else np.asarray(y)
    print("K2 max err / mean|y| =", np.abs(yy - yref).max() / np.abs(yref).mean())

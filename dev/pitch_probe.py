"""dev: do K1 / K2 care about the row pitch of X?  (2048-byte rows put every 128-B column chunk of all rows on the same few memory
channels: partition camping)  usage: python dev/pitch_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
n, d, l = 100000, 512, 74
ctx = petal.Context(0); ctx.set_profiling(2)
g = torch.Generator(device="cuda"); g.manual_seed(1)
p = np.random.default_rng(7).standard_normal((d, l)).astype(np.float32)
for pitch in (512, 528, 544, 576, 640, 520):
    buf = torch.empty((n, pitch), device="cuda")
    x = buf[:, :d]
    x.copy_(torch.randn((n, d), generator=g, device="cuda") * 2 + 0.5)
    mu = x[:4096].mean(0).cpu().numpy().astype(np.float32)
    res = {}
    for name in ("K1", "K2"):
        ms, cnt = 0.0, 0
        for it in range(10):
            if name == "K1": z = petal.gemm_xp(x, p, mu, ctx=ctx); st = ctx.stats(); key = "xp"
            else: y = petal.gemm_atb(x, z, mu, ctx=ctx); st = ctx.stats(); key = "atb"
            if it >= 4: ms += st[key + "_ms"]; cnt += st[key + "_launches"]
        res[name] = ms / max(cnt, 1) * 1e3
    print(f"row pitch {pitch * 4} B: K1 {res['K1']:.1f} us  K2 {res['K2']:.1f} us", flush=True)

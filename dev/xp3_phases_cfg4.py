"""dev: per-chunk phase cycles of k_xp3 / per-stage of k_atb3 at the configs[3] share shape (250000 x 1024, l = 138)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
lib = petal.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpetal_dbg.so"))
ctx = petal.Context(0, lib=lib)
ctx.set_profiling(2)
n, d, l = 250000, 1024, 138
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn((n, d), generator=g, device="cuda") * 2 + 0.5
p = np.random.default_rng(7).standard_normal((d, l)).astype(np.float32)
mu = x[:4096].mean(0).cpu().numpy().astype(np.float32)
cyc = (C.c_longlong * 32)(); dbg = (C.c_int * 4)()
for rep in range(3): z = petal.gemm_xp(x, p, mu, ctx=ctx)
lib.petal_debug_counters(cyc, dbg)
z = petal.gemm_xp(x, p, mu, ctx=ctx); st = ctx.stats()
lib.petal_debug_counters(cyc, dbg)
waves = max(cyc[26], 1); chunks = d // 32
names = ["barrier", "xwait", "split", "issue", "mfma", "pstore"]
per = [cyc[20 + i] / waves / chunks for i in range(6)]
print(f"K1 {st['xp_ms']*1e3:.1f} us, waves {waves}; cycles per chunk per wave:", {k: round(v) for k, v in zip(names, per)}, "sum", round(sum(per)), "(108 MFMAs per chunk)")
zt = torch.zeros((n, 144), device="cuda"); zt[:, :l] = torch.randn((n, l), generator=g, device="cuda")
for rep in range(3): y = petal.gemm_atb(x, zt, mu, ctx=ctx)
lib.petal_debug_counters(cyc, dbg)
y = petal.gemm_atb(x, zt, mu, ctx=ctx); st = ctx.stats()
lib.petal_debug_counters(cyc, dbg)
stages = max(cyc[15], 1)
names = ["barrier", "split", "issue", "mfma", "zstage"]
per = [cyc[10 + i] / stages for i in range(5)]
print(f"K2 {st['atb_ms']*1e3:.1f} us, wave-stages {stages}; cycles per stage per wave:", {k: round(v) for k, v in zip(names, per)}, "sum", round(sum(per)), "(108 MFMAs per stage)")

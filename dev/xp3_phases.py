"""dev: per-chunk phase cycles of k_xp3 (needs dev/libpetal_dbg.so built with -DPETAL_DEBUG_COUNTERS)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
lib = petal.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpetal_dbg.so"))
ctx = petal.Context(0, lib=lib)
ctx.set_profiling(2)
for n in (100000, 1000000):
    d, l = 512, 74
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randn((n, d), generator=g, device="cuda") * 2 + 0.5
    p = np.random.default_rng(7).standard_normal((d, l)).astype(np.float32)
    mu = x[:4096].mean(0).cpu().numpy().astype(np.float32)
    cyc = (C.c_longlong * 32)(); dbg = (C.c_int * 4)()
    for rep in range(3):
        z = petal.gemm_xp(x, p, mu, ctx=ctx)
    lib.petal_debug_counters(cyc, dbg)
    z = petal.gemm_xp(x, p, mu, ctx=ctx)
    st = ctx.stats()
    lib.petal_debug_counters(cyc, dbg)
    waves = max(cyc[26], 1); chunks = 16
    names = ["vmwait", "barrier", "split(+pdma)", "xdma", "mfma", "-"] if os.environ.get("PETAL_XP4", "1") != "0" else ["barrier", "xwait", "split", "issue", "mfma", "pstore"]
    per = [cyc[20 + i] / waves / chunks for i in range(6)]
    print(f"n={n}: K1 {st['xp_ms']*1e3:.1f} us, waves {waves}; cycles per chunk per wave:", {k: round(v) for k, v in zip(names, per)}, "sum", round(sum(per)))
# ---- K2 (k_atb3) phases: cycles per 32-row stage per wave
for n in (100000, 1000000):
    d, l = 512, 74
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randn((n, d), generator=g, device="cuda") * 2 + 0.5
    z = torch.randn((n, 80), generator=g, device="cuda"); z[:, l:] = 0
    mu = x[:4096].mean(0).cpu().numpy().astype(np.float32)
    cyc = (C.c_longlong * 32)(); dbg = (C.c_int * 4)()
    for rep in range(3):
        y = petal.gemm_atb(x, z, mu, ctx=ctx)
    lib.petal_debug_counters(cyc, dbg)
    y = petal.gemm_atb(x, z, mu, ctx=ctx)
    st = ctx.stats()
    lib.petal_debug_counters(cyc, dbg)
    stages = max(cyc[15], 1)
    names = ["barrier", "centre+split", "issue", "mfma", "z split+store"]
    print(f"n={n}: K2 {st['atb_ms']*1e3:.1f} us; cycles per stage per wave:", {k: round(cyc[10 + i] / stages) for i, k in enumerate(names)}, "sum", round(sum(cyc[10 + i] for i in range(5)) / stages))

"""dev: per-stage phase cycles of k_pow3 (needs dev/libpetal_dbg.so built with -DPETAL_DEBUG_COUNTERS: dev/build_dbg.sh)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
lib = petal.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpetal_dbg.so"))
ctx = petal.Context(0, lib=lib)
ctx.set_profiling(2)
names = ["split+P1h0", "write+barA", "add h0", "load+P1h1", "barB+write+barC", "add h1+zB", "barD", "P2"]
for n in (100000, 1000000):
    for N in (74, 64):
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        x = torch.randn((n, 512), generator=g, device="cuda") * 2 + 0.5
        p = np.random.default_rng(7).standard_normal((512, N)).astype(np.float32)
        mu = x[:4096].mean(0).cpu().numpy().astype(np.float32)
        cyc = (C.c_longlong * 32)(); dbg = (C.c_int * 4)()
        for rep in range(3):
            petal.power_pass(x, p, mu, ctx=ctx)
        lib.petal_debug_counters(cyc, dbg)
        petal.power_pass(x, p, mu, ctx=ctx)
        st = ctx.stats()
        lib.petal_debug_counters(cyc, dbg)
        ws = max(cyc[8], 1)     # wave-stages
        per = [cyc[i] / ws for i in range(8)]
        print(f"n={n} N={N}: pass {st['pow_ms']*1e3:.1f} us; cycles per stage per wave:", {k: round(v) for k, v in zip(names, per)}, "sum", round(sum(per)), flush=True)

# absolute timeline of ONE stage of one workgroup (its 8 waves): when each wave passed each mark, relative to the earliest
tr = (C.c_longlong * 128)()
lib.petal_debug_trace(tr)
base = min(tr[w * 16 + 7] for w in range(8)) if any(tr) else 0
# the marks are stamped in the order 0..7 within a stage; mark 7 of the PREVIOUS stage is not recorded, so show marks relative to the earliest mark 0
t0 = min(tr[w * 16 + 0] for w in range(8))
print("one stage of workgroup 7 (cycles after the earliest mark 0):")
print("wave " + " ".join(f"{nm[:10]:>11s}" for nm in names))
for w in range(8):
    print(f"{w:4d} " + " ".join(f"{tr[w * 16 + i] - t0:11d}" for i in range(8)))

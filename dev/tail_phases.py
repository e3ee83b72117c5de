"""Newton-Schulz step counts and phase cycles of the FastICA tail kernel (needs dev/libpetal_dbg.so, -DPETAL_DEBUG_COUNTERS)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_ica
lib = petal.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpetal_dbg.so"))
ctx = petal.Context(0, lib=lib)
cyc = (C.c_longlong * 32)(); dbg = (C.c_int * 4)()
for (n, d, nc, seed) in ((200000, 256, 32, 5), (500000, 512, 64, 8)):
    x = torch.from_numpy(synth_ica(n, d, nc, seed=seed, dtype=np.float32)).cuda()
    w0 = np.random.default_rng(1).standard_normal((nc, nc)).astype(np.float32)
    lib.petal_debug_counters(cyc, dbg)
    m = petal.FastIca(ctx=ctx, n_components=nc).fit(x, w_init=w0)
    lib.petal_debug_counters(cyc, dbg)
    c = [int(v) for v in cyc]
    calls, steps = dbg[2], dbg[1]
    print(f"{n}x{d} nc={nc}: n_iter={m.n_iter}; polar calls {calls}, Newton-Schulz steps {steps} ({steps / max(calls, 1):.1f} per call); "
          f"clock ticks per step (thread 0): T = X X^T {c[17] / max(steps, 1):.0f}, the sum's barrier behind it {c[18] / max(steps, 1):.0f}, the rest of the step {c[16] / max(steps, 1):.0f}")

"""phase cycle counters of the one-workgroup kernels (needs dev/libpetal_dbg.so built with -DPETAL_DEBUG_COUNTERS)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
lib = petal.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpetal_dbg.so"))
ctx = petal.Context(0, lib=lib)
n, d, k = 100000, 512, 64
from synth_data import synth_pca
x = torch.from_numpy(synth_pca(n, d, k, seed=2)).cuda()
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
m = petal.RandomizedPca(k, ctx=ctx, n_iter=5)
cyc = (C.c_longlong * 32)(); dbg = (C.c_int * 4)()
for rep in range(3):
    m.fit(x, omega=om)
    lib.petal_debug_counters(cyc, dbg)
    print("cycles:", [int(v) for v in cyc], "dbg", list(dbg))

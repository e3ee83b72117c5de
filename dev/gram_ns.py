"""dev: the fp64 Gram kernel at 500000 x 512 / 200000 x 256 for a given PETAL_GRAM_NS (row split)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
ctx = petal.Context(0)
for (n, d) in ((500000, 512), (200000, 256)):
    g = torch.Generator(device="cuda"); g.manual_seed(8)
    x = torch.randn((n, d), generator=g, device="cuda", dtype=torch.float32)
    m = petal.Pca(4, ctx=ctx)
    m.fit(x); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): m.fit(x)
    torch.cuda.synchronize()
    print(os.environ.get("PETAL_GRAM_NS", "auto"), n, d, "Pca(4).fit %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)

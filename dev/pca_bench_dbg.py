import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
def run(ctx, x, k, tag):
    m = petal.Pca.new(k, ctx)
    for _ in range(5): m.fit(x)
    t0 = time.perf_counter()
    for _ in range(20): m.fit(x)
    print(tag, f"{(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
n, d, k = 200000, 256, 32
x = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float32)).cuda()
A = petal.Context(0)
run(A, x, k, "ctx A, first")
run(A, x, k, "ctx A, again")
x2 = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float32)).cuda()
run(A, x2, k, "ctx A, second tensor")
B = petal.Context(0)
run(B, x, k, "ctx B")
run(A, x, k, "ctx A after B exists")
del B
run(A, x, k, "ctx A after B deleted")

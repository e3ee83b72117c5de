"""dev: RandomizedPca / FastIca / Pca fits alternating on ONE ctx, shapes changing, results compared with a fresh ctx each time
(the side stream's fork / join and the hot / cold pool lists under reuse)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca, synth_ica
rng = np.random.default_rng(5)
ctx = petal.Context(0)
bad = 0
for rep in range(40):
    kind = rep % 3
    n = int(rng.choice([4096, 20000, 50001])); d = int(rng.choice([64, 256, 512])); k = int(rng.choice([8, 16, 32]))
    fresh = petal.Context(0)
    if kind == 0:
        x = torch.from_numpy(synth_pca(n, d, k, seed=rep, dtype=np.float32)).cuda()
        om = rng.standard_normal((d, k + 10)).astype(np.float32)
        a = petal.RandomizedPca(k, ctx=ctx, n_iter=4).fit(x, omega=om); b = petal.RandomizedPca(k, ctx=fresh, n_iter=4).fit(x, omega=om)
        same = np.array_equal(a.components(), b.components()) and np.array_equal(a.singular_values(), b.singular_values())
    elif kind == 1:
        x = torch.from_numpy(synth_ica(n, d, k, seed=rep, dtype=np.float32)).cuda()
        w0 = rng.standard_normal((k, k)).astype(np.float32)
        a = petal.FastIca(ctx=ctx, n_components=k).fit(x, w_init=w0); b = petal.FastIca(ctx=fresh, n_components=k).fit(x, w_init=w0)
        same = np.array_equal(a.components, b.components) and a.n_iter == b.n_iter
    else:
        x = torch.from_numpy(synth_pca(n, d, k, seed=rep, dtype=np.float32)).cuda()
        a = petal.Pca(k, ctx=ctx).fit(x); b = petal.Pca(k, ctx=fresh).fit(x)
        same = np.array_equal(a.components(), b.components()) and np.array_equal(a.singular_values(), b.singular_values())
    fresh.close()
    if not same:
        bad += 1
        print("MISMATCH", rep, kind, n, d, k, flush=True)
print("alternating fits:", 40 - bad, "of 40 bit-identical to a fresh ctx")

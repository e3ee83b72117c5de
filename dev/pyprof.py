"""dev: where the Python mirror spends its time around petal_rpca_fit (cProfile over 300 fits)"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca
n, d, k = 100000, 512, 64
xd = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float32)).cuda()
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
ctx = petal.Context(0)
m = petal.RandomizedPca(k, ctx=ctx, n_iter=5)
for _ in range(30): m.fit(xd, omega=om)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): m.fit(xd, omega=om)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

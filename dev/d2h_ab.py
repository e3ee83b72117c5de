"""dev: the results' way out -- copy kernel into the pinned ring against hipMemcpyAsync -- two contexts in ONE process, fits alternating"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca, synth_ica
def make(memcpy):
    if memcpy: os.environ["PETAL_D2H_MEMCPY"] = "1"
    else: os.environ.pop("PETAL_D2H_MEMCPY", None)
    return petal.Context(0)
ctxs = {"kernel": make(False), "memcpy": make(True)}
os.environ.pop("PETAL_D2H_MEMCPY", None)
def run(name, fit_of, reps=60):
    models = {k: fit_of(c) for k, c in ctxs.items()}
    for k in models:
        for _ in range(60): models[k]()
    ts = {k: [] for k in models}
    for rep in range(reps):
        for k in models:
            t0 = time.perf_counter(); models[k](); ts[k].append((time.perf_counter() - t0) * 1e6)
    print(name, {k: f"median {np.median(v):.1f} us, min {np.min(v):.1f}" for k, v in ts.items()}, flush=True)
n, d, k, it = 100000, 512, 64, 5
xd = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=np.float32)).cuda()
om = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
def rp(c):
    m = petal.RandomizedPca(k, ctx=c, n_iter=it)
    return lambda: m.fit(xd, omega=om)
run("rpca cfg2", rp); run("rpca cfg2", rp)
xt = torch.from_numpy(synth_pca(200000, 256, 32, seed=5, dtype=np.float32)).cuda()
def pc(c):
    m = petal.Pca(32, ctx=c)
    return lambda: m.fit(xt)
run("pca tall", pc); run("pca tall", pc)
xs = torch.from_numpy(synth_ica(200000, 256, 32, seed=3, dtype=np.float32)).cuda() if hasattr(__import__("synth_data"), "synth_ica") else xt
w0 = np.random.default_rng(1).standard_normal((32, 32))
def ic(c):
    m = petal.FastIca(ctx=c, n_components=32)
    return lambda: m.fit(xs, w_init=w0)
run("fastica cfg3", ic, 30)

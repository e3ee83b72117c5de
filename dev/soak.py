"""dev: repeat the same fits many times and compare the results BITWISE (a race in the barrier / LDS choreography of k_pow3, k_gram5,
k_ica3p or the tail would show as a run that differs).  usage: python dev/soak.py [repeats]"""
import sys, os, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import petal_decomposition_amd as petal
from synth_data import synth_pca, synth_ica
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = petal.Context(0)
def h(*arrs):
    m = hashlib.sha256()
    for a in arrs: m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()[:16]
bad = 0
cases = []
for (n, d, k, it) in ((100000, 512, 64, 5), (33333, 500, 40, 4), (250037, 512, 64, 3), (20000, 1024, 128, 3)):
    x = torch.from_numpy(synth_pca(n, d, k, seed=n % 97, dtype=np.float32)).cuda()
    om = np.random.default_rng(5).standard_normal((d, k + 10)).astype(np.float32)
    cases.append((f"rpca {n}x{d} k={k} n_iter={it}", lambda x=x, om=om, k=k, it=it: (lambda m: h(m.components(), m.singular_values()))(petal.RandomizedPca(k, ctx=ctx, n_iter=it).fit(x, omega=om))))
for (n, d, nc) in ((200000, 256, 32), (60011, 512, 64), (20011, 300, 8), (50000, 1024, 16)):
    x = torch.from_numpy(synth_ica(n, d, nc, seed=n % 89, dtype=np.float32)).cuda()
    w0 = np.random.default_rng(6).standard_normal((nc, nc)).astype(np.float32)
    cases.append((f"ica {n}x{d} nc={nc}", lambda x=x, w0=w0, nc=nc: (lambda m: h(m.components, m.means, np.array([m.n_iter])))(petal.FastIca(ctx=ctx, n_components=nc).fit(x, w_init=w0))))
for name, fn in cases:
    t0 = time.perf_counter()
    ref = fn()
    diff = sum(fn() != ref for _ in range(reps))
    bad += diff
    print(f"{'ok  ' if diff == 0 else 'FAIL'} {name}: {reps} repeats, {diff} differ from the first ({time.perf_counter() - t0:.1f} s)", flush=True)
print("differing runs:", bad)

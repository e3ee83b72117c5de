"""numpy model: what would 16-bit (two bf16 planes) X and z cost in the INTERMEDIATE power iterations of RandomizedPca, the last pass exact?
Component errors against the fp64 oracle with the same Omega, for (a) only P rounded (today), (b) P, X and z rounded in the intermediate passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import petal_oracle as po
import parity_cases as pc

def r16(a):   # round to 16 significant bits (sum of two bf16 pieces, round to nearest): keep 16 bits of the fp32 mantissa
    a = np.asarray(a, dtype=np.float32)
    h = (a.view(np.uint32) + 0x8000 & 0xFFFF0000).view(np.float32) if False else None
    # two-step: h = bf16(a), m = bf16(a - h)
    def bf(x):
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)
    hh = bf(a); mm = bf(a - hh)
    return (hh.astype(np.float64) + mm.astype(np.float64))

def chol_rebase(Y):
    G = Y.T @ Y
    R = np.linalg.cholesky(G).T
    return np.linalg.solve(R.T, Y.T).T

def fit(x, om, k, n_iter, mode):
    xc = x.astype(np.float64) - x.astype(np.float64).mean(axis=0).astype(np.float32).astype(np.float64)
    x2 = r16(xc) if mode == "all" else xc
    # first fused pass: Y' = Xc^T (Xc Omega2)
    P = r16(om)
    npass = n_iter + 1
    for it in range(npass):
        last = it == npass - 1
        if last or mode == "p":
            Z = xc.astype(np.float32).astype(np.float64) @ P
            Y = xc.T @ Z.astype(np.float32).astype(np.float64)
        else:
            Z = r16((x2 @ P).astype(np.float32))
            Y = x2.T @ Z
        if not last:
            P = r16(chol_rebase(Y))
    # QR of Z via its Gram (P^T Y), B = R^-T Y^T, SVD
    G = P.T @ Y
    G = (G + G.T) / 2
    R = np.linalg.cholesky(G).T
    B = np.linalg.solve(R.T, Y.T)
    u, s, vt = np.linalg.svd(B, full_matrices=False)
    return vt[:k], s[:k]

rng = np.random.default_rng(0)
def make(n, d, k, kind):
    if kind == "planted": return po.synth_pca(n, d, k, seed=7, dtype=np.float32)
    if kind in ("geo97", "rsqrt"): return pc.slow_decay_matrix(n, d, kind, 7)
    r = np.random.default_rng(7)
    m = min(n, d)
    if kind.startswith("geo"):
        sv = float(kind[3:]) ** np.arange(m)
    else:   # "noiseX": k planted values 1 .. 0.1 on a flat floor X
        fl = float(kind[5:])
        sv = np.concatenate([np.linspace(1.0, 0.1, k), np.full(m - k, fl * 0.1)])
    u, _ = np.linalg.qr(r.standard_normal((n, m)))
    v, _ = np.linalg.qr(r.standard_normal((d, m)))
    return ((u * (sv * 30.0)) @ v.T + r.standard_normal(d)).astype(np.float32)
cases = [(20000, 512, 64, 5, kd) for kd in (sys.argv[1:] or ["planted", "geo97", "rsqrt"])]
for (n, d, k, n_iter, kind) in cases:
    x = make(n, d, k, kind)
    om = rng.standard_normal((d, k + 10)).astype(np.float32).astype(np.float64)
    o = po.RandomizedPcaOracle(k, n_iter=n_iter)
    o._inner_fit(x.astype(np.float64), omega=om)
    res = {}
    for mode in ("p", "all"):
        c, s = fit(x, om, k, n_iter, mode)
        sg = np.sign(np.sum(c * o.components, axis=1))
        rel = np.linalg.norm(c * sg[:, None] - o.components, axis=1) / np.linalg.norm(o.components, axis=1)
        res[mode] = (rel.max(), np.median(rel), np.abs(s / o.singular - 1).max())
    print(f"{n}x{d} k={k} n_iter={n_iter} {kind:8s}: only P rounded: comp max {res['p'][0]:.2e} (median {res['p'][1]:.1e}), sigma {res['p'][2]:.1e} | P, X, z rounded in the "
          f"intermediate passes: comp max {res['all'][0]:.2e} (median {res['all'][1]:.1e}), sigma {res['all'][2]:.1e}", flush=True)

"""dev: algorithm-level fuzz on the HOST SIMULATION (no GPU): the product's algo.cpp / api.cpp over oracle/cpu_ops.cpp, small shapes, the
conditions round 6's device sweeps found trouble in -- uncentred data far off centre, few iterations with many components, wide matrices,
rank deficiency, both data types and GEMM modes.  usage: python dev/fuzz_hostsim6.py <seed> <cases>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hostsim
import petal_decomposition_amd as petal
import parity_cases as pc
from oracle import petal_oracle as po
ctx = hostsim.context()
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed0)
bad = 0
def report(tag, ok, msg):
    global bad
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} {tag}: {msg}", flush=True)
for case in range(ncase):
    dt = np.float32 if rng.integers(0, 2) else np.float64
    off = float(rng.choice([0.0, 3.0, 40.0, 300.0])); cent = bool(rng.integers(0, 2))
    # exact Pca
    n = int(rng.choice([20, 50, 120, 300, 1500])); d = int(rng.choice([8, 30, 64, 100, 130, 200]))
    k = int(rng.integers(1, max(2, min(n, d, 40))))
    x = po.synth_pca(n, d, k, seed=100 + case, dtype=np.float64)
    x = (x + off * x.std(axis=0) * np.sign(np.random.default_rng(case).standard_normal(d))).astype(dt)
    try:
        o = po.PcaOracle(k, centering=cent, thin=True); o._inner_fit(x.astype(np.float64))
        m = petal.Pca(k, centering=cent, ctx=ctx); m.fit(x)
        rel = pc.rowwise_rel(m.components().astype(np.float64), o.components).max()
        srel = np.abs(np.asarray(m.singular_values(), dtype=np.float64) / o.singular - 1).max()
        kappa = o.singular[0] / max(o.singular[k - 1], 1e-300)
        tol = (5e-5 if dt == np.float32 else 1e-8) * max(1.0, kappa / 1e3)
        e32 = None
        if (rel > tol or srel > tol) and dt == np.float32:
            o32 = po.PcaOracle(k, centering=cent, thin=True); o32._inner_fit(x)
            e32 = pc.rowwise_rel(o32.components.astype(np.float64), o.components).max()
        ok = (rel <= tol and srel <= tol) or (e32 is not None and e32 >= 0.5 * rel and srel <= 10 * tol)
        report(f"pca {dt.__name__} n={n} d={d} k={k} off={off} cent={cent}", ok, f"rel {rel:.1e} sigma {srel:.1e} kappa {kappa:.1e} tol {tol:.1e} fp32-oracle {e32}")
    except Exception as e:
        report(f"pca {dt.__name__} n={n} d={d} k={k} off={off} cent={cent}", False, str(e)[:200])
    # RandomizedPca
    n = int(rng.choice([60, 300, 1500, 3000])); d = int(rng.choice([24, 64, 100, 160]))
    k = int(rng.integers(1, max(2, min(n, d) - 11))); it = int(rng.choice([0, 1, 2, 3, 4, 7]))
    mode = "bf16x3" if rng.integers(0, 2) else "fp32"
    ctx.set_gemm_mode(mode)
    x = po.synth_pca(n, d, k, seed=200 + case, dtype=np.float64)
    x = (x + off * x.std(axis=0) * np.sign(np.random.default_rng(case).standard_normal(d))).astype(dt)
    om = np.random.default_rng(300 + case).standard_normal((d, k + 10)).astype(dt)
    try:
        o = po.RandomizedPcaOracle(k, centering=cent, n_iter=it); o._inner_fit(x.astype(np.float64), omega=om.astype(np.float64))
        m = petal.RandomizedPca(k, centering=cent, ctx=ctx, n_iter=it).fit(x, omega=om)
        st = ctx.stats()
        rel = pc.rowwise_rel(m.components().astype(np.float64), o.components).max()
        srel = np.abs(np.asarray(m.singular_values(), dtype=np.float64) / np.maximum(o.singular, 1e-300) - 1).max()
        tol = max(2e-5, 3e-6 / (1.0 - 10.0 ** (-3.0 / max(k, 1)))) if dt == np.float32 else 1e-8
        e32 = None
        if (rel > tol or srel > max(tol, 5e-5)) and dt == np.float32:
            o32 = po.RandomizedPcaOracle(k, centering=cent, n_iter=it); o32._inner_fit(x, omega=om)
            e32 = pc.rowwise_rel(o32.components.astype(np.float64), o.components).max()
        ok = (rel <= tol and srel <= max(tol, 5e-5)) or (e32 is not None and e32 >= 0.5 * rel)
        report(f"rpca {dt.__name__} {mode} n={n} d={d} k={k} it={it} off={off} cent={cent}", ok,
               f"rel {rel:.1e} sigma {srel:.1e} tol {tol:.1e} fp32-oracle {e32} redo {st['rpca_redo']} eigh_redo {st['eigh_redo']}")
    except Exception as e:
        report(f"rpca {dt.__name__} {mode} n={n} d={d} k={k} it={it} off={off} cent={cent}", False, str(e)[:200])
    ctx.set_gemm_mode("fp32")
print("failures:", bad)
# FastICA, strictly: the oracle started from the LIBRARY's side of the whitening rows' sign ambiguity (w_init . diag(s), s_j = the sign the
# library's convention -- largest-magnitude component positive -- gives LAPACK's row j), so both run the same trajectory
bad = 0
rng = np.random.default_rng(seed0 + 1000)
for case in range(ncase):
    dt = np.float32 if rng.integers(0, 2) else np.float64
    n = int(rng.choice([500, 2000, 5000])); d = int(rng.choice([4, 8, 16, 40, 100, 130]))
    nc = int(rng.integers(2, min(d, 24) + 1)); off = float(rng.choice([0.0, 3.0, 40.0]))
    x = po.synth_ica(n, d, nc, seed=400 + case, dtype=np.float64)
    x = (x + off * x.std(axis=0) * np.sign(np.random.default_rng(case).standard_normal(d))).astype(dt)
    w0 = np.random.default_rng(500 + case).standard_normal((nc, nc))
    tag = f"ica {dt.__name__} n={n} d={d} nc={nc} off={off}"
    try:
        o = po.FastIcaOracle(n_components=nc, whiten="eigh"); o.fit(x.astype(np.float64), w_init=w0)
        kk = o.k_
        s = np.sign(kk[np.arange(nc), np.abs(kk).argmax(axis=1)])
        o2 = po.FastIcaOracle(n_components=nc, whiten="eigh"); o2.fit(x.astype(np.float64), w_init=w0 * s[None, :]); yo = o2.transform(x.astype(np.float64))
        m = petal.FastIca(ctx=ctx, n_components=nc); y = np.asarray(m.fit_transform(x, w_init=w0.astype(dt)), dtype=np.float64)
        c = np.abs(y.T @ yo); perm = c.argmax(axis=1)
        dev = max(np.abs(1.0 - c[np.arange(nc), perm]).max(), np.abs(c - np.eye(nc)[perm]).max()) if sorted(perm.tolist()) == list(range(nc)) else 9.0
        tol = 5e-3 if dt == np.float32 else 1e-6
        report(tag, dev <= tol and abs(m.n_iter - o2.n_iter) <= 1, f"dev {dev:.1e} iterations {m.n_iter}/{o2.n_iter} (plain oracle {o.n_iter}) redo {ctx.stats()['ica_redo']}")
    except Exception as e:
        report(tag, False, str(e)[:200])
print("ica failures:", bad)

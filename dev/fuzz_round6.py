"""dev: random shapes on the round-6 paths against the oracle -- RandomizedPca whose re-basing / final stage run on k_chol_rt4 + the
triangular solves in EVERY data type and GEMM mode (any l <= 140), the fused pass at 241 .. 256 features beside 497 .. 512, short and long
iteration counts, centred and not; FastICA / exact Pca whose whitening eigen-solve orthonormalises through the same kernels (few components of
many features).  usage: python dev/fuzz_round6.py <seed> <cases>   (FUZZ_GEMM=fp32 for the fp32-MFMA mode)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import petal_decomposition_amd as petal
import parity_cases as pc
ctx = petal.Context(0)
if os.environ.get("FUZZ_GEMM"): ctx.set_gemm_mode(os.environ["FUZZ_GEMM"])
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed0)
only = os.environ.get('FUZZ6_ONLY', '')
bad = 0
from oracle import petal_oracle as po
def tol_for(k):   # (dev/fuzz_rpca.py's: the planted spectrum spans three decades over k components)
    return max(2e-5, 3e-6 / (1.0 - 10.0 ** (-3.0 / max(k, 1))))
def classify(n, d, k, it, seed, cent, err_text):
    """a miss of the tolerance: the ORACLE in float32 on the same input and Omega (the reference's own arithmetic) against the fp64 oracle"""
    x = po.synth_pca(n, d, k, seed=seed, dtype=np.float32)
    om = np.random.default_rng(seed + 1000).standard_normal((d, k + 10)).astype(np.float32)
    o = po.RandomizedPcaOracle(k, centering=cent, n_iter=it); o._inner_fit(x.astype(np.float64), omega=om.astype(np.float64))
    o32 = po.RandomizedPcaOracle(k, centering=cent, n_iter=it); o32._inner_fit(x, omega=om)
    e32 = pc.rowwise_rel(o32.components.astype(np.float64), o.components).max()
    return f"fp32 oracle misses the fp64 one by {e32:.2e}"
def run(tag, fn):
    global bad
    try:
        r = fn()
        print(f"ok   {tag} -> {r}", flush=True)
    except Exception as e:
        bad += 1
        print(f"FAIL {tag}: {str(e)[:200]}", flush=True)
for case in range(ncase):
    # 1. the fused pass at both feature widths (fp32 data, default mode) / K1 + K2 otherwise
    d = int(rng.choice([241, 250, 255, 256, 497, 500, 511, 512]))
    n = int(rng.choice([4096, 8192, 10000, 20011, 33333, 60000]))
    k = int(rng.choice([3, 16, 33, 54, 64, 70]))
    it = int(rng.choice([2, 3, 4, 5, 7]))
    cent = bool(rng.integers(0, 4))
    dev = bool(rng.integers(0, 2))
    def fused():
        try:
            r = pc.rpca_parity(ctx, n, d, k, it, seed=6000 + case, dtype=np.float32, tol=tol_for(k), tol_sigma=5e-5, centering=cent, device=dev)
        except AssertionError as e:
            raise AssertionError(f"{str(e)[:90]} | {classify(n, d, k, it, 6000 + case, cent, str(e))}")
        st = getattr(pc.rpca_parity, 'last_fit_stats', None) or ctx.stats()
        return f"{r} fused launches {st['pow_launches']} redo {st['rpca_redo']}"
    if only in ("", "fused"): run(f"rpca float32 n={n} d={d} k={k} n_iter={it} cent={cent} dev={dev}", fused)
    # 2. the factorisation orders 1 .. 9 blocks in fp64 (no operand planes anywhere: k_trsm_pack / k_trsm_left_pack write fp64 only)
    d2 = int(rng.choice([64, 100, 160, 200, 320]))
    n2 = int(rng.choice([1000, 3001, 8000]))
    k2 = int(rng.integers(1, min(d2, 130) - 10))
    it2 = int(rng.choice([1, 2, 4, 7]))
    if only in ("", "f64"): run(f"rpca float64 n={n2} d={d2} k={k2} n_iter={it2}",
        lambda: pc.rpca_parity(ctx, n2, d2, k2, it2, seed=7000 + case, dtype=np.float64, tol=1e-9))
    # 3. the same orders for fp32 data off the fused pass's widths
    d3 = int(rng.choice([96, 128, 200, 300, 384, 640, 1024]))
    k3 = int(rng.integers(1, min(d3 // 2, 128)))
    it3 = int(rng.choice([3, 4, 5, 7]))
    def wide():
        try:
            return pc.rpca_parity(ctx, n, d3, k3, it3, seed=8000 + case, dtype=np.float32, tol=tol_for(k3), tol_sigma=5e-5)
        except AssertionError as e:
            raise AssertionError(f"{str(e)[:90]} | {classify(n, d3, k3, it3, 8000 + case, True, str(e))}")
    if only in ("", "wide"): run(f"rpca float32 n={n} d={d3} k={k3} n_iter={it3}", wide)
    # 4. FastICA: few components of many features (subspace iteration with the RT-form orthonormalisation)
    d4 = int(rng.choice([128, 256, 300, 512]))
    nc = int(rng.choice([2, 5, 8, 16, 24, 32]))
    n4 = int(rng.choice([5000, 20000, 50001]))
    dt4 = np.float32 if rng.integers(0, 2) else np.float64
    def ica_strict():
        # STRICT: the oracle is started from the library's side of the whitening rows' sign ambiguity (w_init . diag(s), s_j = the sign the
        # library's convention -- largest-magnitude component positive -- gives LAPACK's row j): same trajectory, same stop
        x = po.synth_ica(n4, d4, nc, seed=9000 + case, dtype=dt4)
        w0 = np.random.default_rng(9000 + case + 7).standard_normal((nc, nc))
        o = po.FastIcaOracle(n_components=nc, whiten="eigh"); o.fit(x.astype(np.float64), w_init=w0)
        sg = np.sign(o.k_[np.arange(nc), np.abs(o.k_).argmax(axis=1)])
        o2 = po.FastIcaOracle(n_components=nc, whiten="eigh"); o2.fit(x.astype(np.float64), w_init=w0 * sg[None, :]); yo = o2.transform(x.astype(np.float64))
        m = petal.FastIca(ctx=ctx, n_components=nc); y = np.asarray(m.fit_transform(x, w_init=w0.astype(dt4)), dtype=np.float64)
        c = np.abs(y.T @ yo); perm = c.argmax(axis=1)
        assert sorted(perm.tolist()) == list(range(nc)), perm
        dev = max(np.abs(1.0 - c[np.arange(nc), perm]).max(), np.abs(c - np.eye(nc)[perm]).max())
        assert dev <= (5e-3 if dt4 == np.float32 else 1e-6) and abs(m.n_iter - o2.n_iter) <= 1, (dev, m.n_iter, o2.n_iter)
        return f"dev {dev:.1e} iterations {m.n_iter}/{o2.n_iter} (plain oracle {o.n_iter})"
    if only in ("", "ica"): run(f"ica {dt4.__name__} n={n4} d={d4} nc={nc}", ica_strict)
    # 5. data far off centre, centred and NOT (uncentred: the mean direction is sigma_1, hundreds of times the planted spectrum's head)
    d5 = int(rng.choice([256, 300, 512, 1024]))
    k5 = int(rng.choice([8, 32, 64, 100]))
    it5 = int(rng.choice([3, 4, 5, 7]))
    off = float(rng.choice([3.0, 40.0, 300.0]))
    cent5 = bool(rng.integers(0, 2))
    def offcentre():
        x = po.synth_pca(n, d5, k5, seed=9500 + case, dtype=np.float64)
        x = (x + off * x.std(axis=0) * np.sign(np.random.default_rng(case).standard_normal(d5))).astype(np.float32)
        om = np.random.default_rng(9500 + case + 1000).standard_normal((d5, k5 + 10)).astype(np.float32)
        try:
            r = pc.rpca_parity(ctx, n, d5, k5, it5, seed=9500 + case, dtype=np.float32, tol=tol_for(k5), tol_sigma=5e-5, centering=cent5, x=x)
        except AssertionError as e:
            o = po.RandomizedPcaOracle(k5, centering=cent5, n_iter=it5); o._inner_fit(x.astype(np.float64), omega=om.astype(np.float64))
            o32 = po.RandomizedPcaOracle(k5, centering=cent5, n_iter=it5); o32._inner_fit(x, omega=om)
            e32 = pc.rowwise_rel(o32.components.astype(np.float64), o.components).max()
            raise AssertionError(f"{str(e)[:90]} | fp32 oracle misses the fp64 one by {e32:.2e}")
        return f"{r} redo {getattr(pc.rpca_parity, 'last_fit_stats', {}).get('rpca_redo')}"
    if only in ("", "off"): run(f"rpca float32 n={n} d={d5} k={k5} n_iter={it5} off={off} cent={cent5}", offcentre)
    # 6. exact Pca (full SVD) of off-centre data, centred and not, both data types, tall and wide
    n6 = int(rng.choice([50, 300, 3001, 20000])); d6 = int(rng.choice([16, 64, 100, 256, 300]))
    k6 = int(rng.integers(1, max(2, min(n6, d6, 48))))
    dt6 = np.float32 if rng.integers(0, 2) else np.float64
    off6 = float(rng.choice([0.0, 3.0, 40.0])); cent6 = bool(rng.integers(0, 2))
    def exact_pca():
        x = po.synth_pca(n6, d6, k6, seed=9700 + case, dtype=np.float64)
        x = (x + off6 * x.std(axis=0) * np.sign(np.random.default_rng(case).standard_normal(d6))).astype(dt6)
        o = po.PcaOracle(k6, centering=cent6, thin=True); o._inner_fit(x.astype(np.float64))
        m = petal.Pca(k6, centering=cent6, ctx=ctx); m.fit(x)
        rel = pc.rowwise_rel(m.components().astype(np.float64), o.components).max()
        srel = np.abs(np.asarray(m.singular_values(), dtype=np.float64) / o.singular - 1).max()
        tol = (5e-5 if dt6 == np.float32 else 1e-8) * (1.0 if (cent6 or off6 == 0.0) else (1.0 + off6))
        if rel > tol or srel > tol:
            e32 = float("nan")
            if dt6 == np.float32:
                o32 = po.PcaOracle(k6, centering=cent6, thin=True); o32._inner_fit(x)
                e32 = pc.rowwise_rel(o32.components.astype(np.float64), o.components).max()
            raise AssertionError(f"components {rel:.2e} sigma {srel:.2e} > {tol:.1e} | fp32 oracle misses the fp64 one by {e32:.2e}")
        return f"rel {rel:.1e} sigma {srel:.1e}"
    if only in ("", "pca"): run(f"pca {dt6.__name__} n={n6} d={d6} k={k6} off={off6} cent={cent6}", exact_pca)
print("failures:", bad)

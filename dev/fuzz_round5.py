"""dev: random shapes on the round-5 paths against the oracle -- FastICA whose covariance and means come from k_gram5 (n >= 4096, 256+ padded
features, ragged last stage / last panel, data off centre), RandomizedPca on the fused pass (497 .. 512 features, any k with l <= 80, n_iter 3 ..
7, the means folded in from PETAL_MEANS_FOLD_ROWS rows).  usage: python dev/fuzz_round5.py <seed> <cases>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import petal_decomposition_amd as petal
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import petal_oracle as po
import parity_cases as pc
os.environ.setdefault("PETAL_MEANS_FOLD_ROWS", "20000")
ctx = petal.Context(0)
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed0)
bad = 0
for case in range(ncase):
    try:
        # FastICA
        n = int(rng.choice([4096, 5000, 8191, 12345, 20011, 40000]))
        d = int(rng.choice([241, 256, 257, 300, 384, 500, 512, 513, 700, 1024]))
        nc = int(rng.choice([2, 5, 8, 16, 24, 32]))
        off = float(rng.choice([0.0, 3.0, 40.0, 300.0]))
        x = po.synth_ica(n, d, nc, seed=1000 + case, dtype=np.float64)
        x = (x + off * x.std(axis=0) * np.sign(rng.standard_normal(d))).astype(np.float32)
        w0 = rng.standard_normal((nc, nc)).astype(np.float32)
        m = petal.FastIca(ctx=ctx, n_components=nc)
        y = np.asarray(m.fit_transform(x, w_init=w0))
        st = ctx.stats()
        mu = x.astype(np.float64).mean(axis=0)
        emu = np.abs(np.asarray(m.means, dtype=np.float64) - mu).max() / max(np.abs(mu).max(), np.abs(x).std())
        o = po.FastIcaOracle(n_components=nc, whiten="eigh")
        o.fit(x.astype(np.float64), w_init=w0.astype(np.float64))
        yo = o.transform(x.astype(np.float64))
        c = np.abs(y.astype(np.float64).T @ yo)
        perm = c.argmax(axis=1)
        dev = np.abs(1.0 - c[np.arange(nc), perm]).max()
        okp = sorted(perm.tolist()) == list(range(nc))
        flag = "ok  " if (okp and dev <= 5e-3 and emu <= 3e-7) else "FAIL"
        bad += flag == "FAIL"
        print(f"{flag} ica n={n} d={d} nc={nc} off={off}: dev {dev:.2e}, means {emu:.1e}, split={st['ica_gram_split']} redo={st['ica_redo']} iters {m.n_iter}/{o.n_iter}", flush=True)
    except Exception as e:
        bad += 1
        print(f"FAIL ica case {case}: {str(e)[:200]}", flush=True)
    try:
        n = int(rng.choice([8192, 10000, 20000, 33333, 60000]))
        d = int(rng.choice([497, 500, 511, 512]))
        k = int(rng.choice([3, 16, 33, 54, 64, 70]))
        it = int(rng.choice([3, 4, 5, 7]))
        r = pc.rpca_parity(ctx, n, d, k, it, seed=2000 + case, dtype=np.float32, tol=2e-5, tol_sigma=2e-5)
        st = ctx.stats()
        print(f"ok   rpca n={n} d={d} k={k} n_iter={it}: {r}, fused launches {st['pow_launches']}, redo {st['rpca_redo']}", flush=True)
    except Exception as e:
        bad += 1
        print(f"FAIL rpca n={n} d={d} k={k} n_iter={it}: {str(e)[:200]}", flush=True)
print("failures:", bad)

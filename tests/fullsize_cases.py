"""BASELINE configs[3] and configs[4] AT THEIR STATED SIZE -- RandomizedPca k = 128 on 2 000 000 x 1024 fp32 and FastIca with 64
components on 4 000 000 x 512 fp32, each sample-sharded 8 ways -- generated block by block straight into HBM.

Shared by the rank workers (tests/fullsize_worker.py: one row block each) and the parent test (tests/test_gpu_fullsize.py:
the single-process fit of the whole 8.2 GB matrix).  Every row block comes from its own seeded device generator, so any
process regenerates any block; the small planted factors (V, s, mu / the mixing matrix) come from numpy seeds and are shared
by all blocks.  The models are the ones of synth_data.synth_pca / synth_ica."""
import numpy as np

WORLD = 8
CFG4 = dict(n=2_000_000, d=1024, k=128, n_iter=7, seed=4)       # configs[3]
CFG5 = dict(n=4_000_000, d=512, nc=64, seed=8, tol=1e-4)        # configs[4]
CHUNK = 50_000


def cfg4_factors():
    c = CFG4
    rng = np.random.default_rng(c["seed"])
    r = 2 * c["k"]
    rho = 10.0 ** (-3.0 / c["k"])
    v, _ = np.linalg.qr(rng.standard_normal((c["d"], r)))
    s = 100.0 * np.sqrt(c["n"]) * rho ** np.arange(r)
    mu = rng.standard_normal(c["d"])
    return v, s, mu


def cfg4_omega():
    return np.random.default_rng(3).standard_normal((CFG4["d"], CFG4["k"] + 10)).astype(np.float32)


def cfg4_block(b, out=None, device="cuda"):
    """rows [b n / 8, (b + 1) n / 8) of the planted 2 000 000 x 1024 matrix (fp32, device); `out`: a view to fill"""
    import torch
    c = CFG4
    rows = c["n"] // WORLD
    v, s, mu = cfg4_factors()
    vt = torch.from_numpy((v * s).T.astype(np.float32)).to(device)            # diag(s) V^T  (r x d)
    mut = torch.from_numpy(mu.astype(np.float32)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(40_000 + b)
    x = out if out is not None else torch.empty((rows, c["d"]), dtype=torch.float32, device=device)
    inv = 1.0 / np.sqrt(c["n"])
    for i in range(0, rows, CHUNK):
        m = min(CHUNK, rows - i)
        gg = torch.randn((m, vt.shape[0]), generator=g, device=device, dtype=torch.float32) * inv
        nz = torch.randn((m, c["d"]), generator=g, device=device, dtype=torch.float32)
        x[i:i + m] = gg @ vt + 0.01 * nz + mut
    return x


def cfg5_mixing():
    return np.random.default_rng(CFG5["seed"]).standard_normal((CFG5["nc"], CFG5["d"])).astype(np.float32)


def cfg5_w0():
    return np.random.default_rng(7).standard_normal((CFG5["nc"], CFG5["nc"])).astype(np.float32)


def cfg5_block(b, out=None, device="cuda", want_sources=False):
    """rows of the 4 000 000 x 512 FastICA matrix: Laplace(0, 1) sources (by inversion) through the mixing matrix + 0.01 N"""
    import torch
    c = CFG5
    rows = c["n"] // WORLD
    a = torch.from_numpy(cfg5_mixing()).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(50_000 + b)
    x = out if out is not None else torch.empty((rows, c["d"]), dtype=torch.float32, device=device)
    src = torch.empty((rows, c["nc"]), dtype=torch.float32, device=device) if want_sources else None
    for i in range(0, rows, CHUNK):
        m = min(CHUNK, rows - i)
        u = torch.rand((m, c["nc"]), generator=g, device=device, dtype=torch.float32) - 0.5
        s = -torch.sign(u) * torch.log1p(-2.0 * u.abs().clamp(max=0.4999999))
        nz = torch.randn((m, c["d"]), generator=g, device=device, dtype=torch.float32)
        x[i:i + m] = s @ a + 0.01 * nz
        if want_sources:
            src[i:i + m] = s
    return (x, src) if want_sources else x


def source_match(y, src):
    """|correlation| matrix (estimated components x planted sources) of standardised columns, on the device"""
    n = y.shape[0]
    ys = (y - y.mean(0)) / y.std(0)
    ss = (src - src.mean(0)) / src.std(0)
    return (ys.T @ ss / n).abs().cpu().numpy()

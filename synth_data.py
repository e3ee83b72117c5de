"""Seeded synthetic inputs of BASELINE.md section 3 / SURVEY.md 8(d) (shared by bench.py, the tests and the
oracle; numpy only, except synth_pca_device which fills a torch tensor on the GPU)."""
import numpy as np

DEVICE_BLOCK = 62_500   # rows per seeded generator block of synth_pca_device (2 000 000 = 32 blocks: any 1/2/4/8/16/32-way split)


def synth_pca_factors(n, d, k, seed):
    """(V d x r orthonormal, s, mu) of the planted model of synth_pca for an n-row matrix"""
    rng = np.random.default_rng(seed)
    r = min(2 * k, d, n)
    rho = 10.0 ** (-3.0 / max(k, 1))
    v, _ = np.linalg.qr(rng.standard_normal((d, r)))
    s = 100.0 * np.sqrt(n) * rho ** np.arange(r)
    mu = rng.standard_normal(d)
    return v, s, mu


def synth_pca_device(n_total, d, k, seed, row_begin, row_end, device, noise=0.01):
    """Rows [row_begin, row_end) of the planted n_total x d fp32 matrix of synth_pca's model, generated on the GPU.

    The matrix is DEFINED block by block (DEVICE_BLOCK rows, each block from its own seeded device generator), so any rank
    of any world size regenerates exactly its rows of the same global matrix: a strong-scaling run at N = 1, 2, 4, 8 factors
    one and the same matrix.  (Not the same numbers as the host generator synth_pca: a different random stream.)"""
    import torch
    v, s, mu = synth_pca_factors(n_total, d, k, seed)
    vt = torch.from_numpy((v * s).T.astype(np.float32)).to(device)   # diag(s) V^T  (r x d)
    mut = torch.from_numpy(mu.astype(np.float32)).to(device)
    x = torch.empty((row_end - row_begin, d), dtype=torch.float32, device=device)
    inv = 1.0 / np.sqrt(n_total)
    g = torch.Generator(device=device)
    b0, b1 = row_begin // DEVICE_BLOCK, (row_end + DEVICE_BLOCK - 1) // DEVICE_BLOCK
    for b in range(b0, b1):
        lo, hi = b * DEVICE_BLOCK, min((b + 1) * DEVICE_BLOCK, n_total)
        g.manual_seed(seed * 1_000_003 + b)
        gg = torch.randn((hi - lo, vt.shape[0]), generator=g, device=device, dtype=torch.float32) * inv
        blk = torch.addmm(mut.expand(hi - lo, d), gg, vt)
        del gg
        blk.add_(torch.randn((hi - lo, d), generator=g, device=device, dtype=torch.float32), alpha=noise)
        a, e = max(lo, row_begin), min(hi, row_end)
        x[a - row_begin:e - row_begin] = blk[a - lo:e - lo]
        del blk
    return x


def synth_pca(n, d, k, seed, dtype=np.float32, noise=0.01, row_seed=None):
    """Planted low-rank + noise + means:  X = (G diag(s)) V^T + noise*N + 1 mu^T,  G n x r iid N(0,1)/sqrt(n),
    r = 2k, V d x r orthonormal, s_i = 100 sqrt(n) rho^i, rho = 10^(-3/k)  (sigma_1 / sigma_k = 1e3).
    row_seed (sample-sharded runs): V and mu come from `seed` (shared by all row blocks), the rows from `row_seed`."""
    rng = np.random.default_rng(seed)
    r = min(2 * k, d, n)
    rho = 10.0 ** (-3.0 / max(k, 1))
    v, _ = np.linalg.qr(rng.standard_normal((d, r)))
    s = 100.0 * np.sqrt(n) * rho ** np.arange(r)
    mu = rng.standard_normal(d)
    if row_seed is not None:
        rng = np.random.default_rng(row_seed)
    x = np.empty((n, d), dtype=dtype)
    step = 65536
    for i in range(0, n, step):
        m = min(step, n - i)
        g = rng.standard_normal((m, r)) / np.sqrt(n)
        x[i:i + m] = ((g * s) @ v.T + noise * rng.standard_normal((m, d)) + mu).astype(dtype)
    return x


def synth_ica(n, d, nc, seed, dtype=np.float32, noise=0.01, row_seed=None):
    """Laplace sources through a Gaussian mixing matrix + noise:  X = S A + noise*N.
    row_seed (sample-sharded runs): A comes from `seed` (shared by all row blocks), the rows from `row_seed`."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((nc, d))
    if row_seed is not None:
        rng = np.random.default_rng(row_seed)
    x = np.empty((n, d), dtype=dtype)
    step = 65536
    for i in range(0, n, step):
        m = min(step, n - i)
        s = rng.laplace(size=(m, nc))
        x[i:i + m] = (s @ a + noise * rng.standard_normal((m, d))).astype(dtype)
    return x

"""Seeded synthetic inputs of BASELINE.md section 3 / SURVEY.md 8(d) (shared by bench.py, the tests and the
oracle; numpy only)."""
import numpy as np


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

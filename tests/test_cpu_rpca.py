"""The threaded C++ restatement (oracle/cpu_rpca.cpp: own GEMM, LU, Householder QR, Jacobi SVD -- no BLAS / LAPACK) against
the numpy + LAPACK oracle, which the reference's own known-answer tests pin: same X, same Omega, same n_iter."""
import numpy as np
import pytest

from oracle import cpu_rpca
from oracle import petal_oracle as po


def rowwise_rel(a, b):
    s = np.sign(np.sum(a * b, axis=1))
    s[s == 0] = 1
    return np.linalg.norm(a * s[:, None] - b, axis=1) / np.linalg.norm(b, axis=1)


@pytest.mark.parametrize("n,d,k,n_iter,dtype,tol", [(3000, 64, 6, 7, np.float64, 1e-9), (1501, 50, 5, 4, np.float64, 1e-9),
                                                    (4000, 96, 8, 5, np.float32, 2e-4), (300, 40, 12, 0, np.float64, 1e-9)])
def test_cpp_restatement_matches_the_lapack_oracle(n, d, k, n_iter, dtype, tol):
    x = po.synth_pca(n, d, k, seed=n % 89, dtype=dtype)
    om = np.random.default_rng(n + 1).standard_normal((d, k + 10))
    o = po.RandomizedPcaOracle(k, n_iter=n_iter).fit(x.astype(np.float64), omega=om)
    m = cpu_rpca.RandomizedPcaCpp(k, n_iter=n_iter, threads=4).fit(x, om)
    assert rowwise_rel(m.components, o.components).max() <= tol
    assert np.allclose(m.singular, o.singular, rtol=tol)
    assert np.allclose(m.means, o.means, atol=1e-6 * max(1.0, np.abs(o.means).max()))
    assert np.allclose(m.explained_variance_ratio(), o.explained_variance_ratio(), rtol=10 * tol)
    # signs: svd_flip on the same U columns (compare where the deciding |u| is not a near-tie)
    assert np.mean(np.sign(np.sum(m.components * o.components, axis=1)) == 1) >= 0.8


def test_reference_kat_through_the_cpp_restatement(kats):
    c = kats["randomized_pca"]          # src/pca.rs:949-970: rank-1 3 x 2 data, any Omega
    x = np.array(c["x"], dtype=np.float64)
    om = np.random.default_rng(0).standard_normal((2, 11))
    m = cpu_rpca.RandomizedPcaCpp(1, threads=1).fit(x, om)
    y = (x - m.means) @ m.components.T
    assert np.allclose(np.abs(y[:, 0]), c["abs_y"], atol=c["tol"], rtol=0)
    assert np.allclose(y @ m.components + m.means, x, atol=c["tol"], rtol=0)

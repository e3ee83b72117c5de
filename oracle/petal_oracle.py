"""CPU restatement of petal-decomposition's dense hot path.  TEST INFRASTRUCTURE ONLY.

This module is the parity *oracle*: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``petal-decomposition_amd``) never imports, links or executes anything in ``oracle/``.

It restates the reference algorithm (petabi/petal-decomposition v0.9.0) line by line on
numpy + the same LAPACK routines the crate calls (through ``scipy.linalg.lapack``):

    reference                                   here
    ---------------------------------------     ---------------------------------
    linalg::svd    -> gesvd    (linalg.rs:70)   lapack_svd_full / lapack_svd_left
    linalg::svddc  -> gesdd 'S'(linalg.rs:101)  lapack_svddc
    linalg::qr     -> gelqf+unglq (linalg.rs:127) lapack_qr_thin (geqrf+orgqr: same reflectors)
    linalg::eigh   -> heev 'V','L' (linalg.rs:39) lapack_eigh (returns the buffer *as the crate reads it*)
    lair lu::Factorized::into_pl (pca.rs:709)   lu_pl (getrf partial pivoting, P.L)

Pinning status: every known-answer test the reference holds for this path
(values transcribed in tests/golden/reference_kats.json) is checked by
tests/test_oracle_golden.py.  Two pieces of third-party arithmetic are NOT under
/root/reference and stay "parity unpinned": the pivoted-LU output of ``lair 0.8`` (any
column re-basing yields the same final subspace; pinned only end-to-end) and the
``rand_pcg``/``rand_distr`` Omega / w_init stream (callers pass Omega / w_init explicitly).
The reference itself (Rust) cannot be built in this image (no cargo/rustc).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import lapack as _lp

__all__ = [
    "svd_flip", "lu_pl", "lapack_qr_thin", "lapack_svddc", "lapack_svd_full", "lapack_eigh",
    "randomized_range_finder", "randomized_svd", "RandomizedPcaOracle", "PcaOracle",
    "transform", "inverse_transform", "transform_with_u",
    "logcosh", "symmetric_decorrelation", "ica_par", "FastIcaOracle",
]


class InvalidInput(ValueError):
    """DecompositionError::InvalidInput (lib.rs:24)."""


class LinalgError(RuntimeError):
    """DecompositionError::LinalgError (lib.rs:26)."""


def _prefix(dtype) -> str:
    return {np.dtype(np.float32): "s", np.dtype(np.float64): "d"}[np.dtype(dtype)]


# --------------------------------------------------------------------------- linalg.rs
def lapack_svd_full(a: np.ndarray):
    """linalg::svd(a, calc_vt=true) -> gesvd('A','A')  (linalg.rs:70-91, lapack.rs:103-132).
    Returns (U m x m, sigma min(m,n), Vt n x n)."""
    f = getattr(_lp, _prefix(a.dtype) + "gesvd")
    u, s, vt, info = f(np.asarray(a, order="F"), compute_uv=1, full_matrices=1)
    if info != 0:
        raise LinalgError("did not converge")
    return np.ascontiguousarray(u), s, np.ascontiguousarray(vt)


def lapack_svd_left(a: np.ndarray):
    """linalg::svd(a, calc_vt=false): the crate asks LAPACK for jobu='N', jobvt='A' on the
    transposed view, i.e. it returns the full LEFT factor U (m x m) and sigma (ica.rs:189)."""
    u, s, _ = lapack_svd_full(a)
    return u, s


def lapack_svddc(a: np.ndarray):
    """linalg::svddc -> gesdd('S')  (linalg.rs:101-122).  Returns (U m x k, sigma k, Vt k x n)."""
    f = getattr(_lp, _prefix(a.dtype) + "gesdd")
    u, s, vt, info = f(np.asarray(a, order="F"), compute_uv=1, full_matrices=0)
    if info != 0:
        raise LinalgError("did not converge")
    return np.ascontiguousarray(u), s, np.ascontiguousarray(vt)


def lapack_qr_thin(a: np.ndarray):
    """linalg::qr -> gelqf + unglq on the transposed view (linalg.rs:127-147).  The LQ of A^T uses
    the same Householder reflectors as the QR of A, so geqrf + orgqr yields the same thin Q."""
    p = _prefix(a.dtype)
    m, n = a.shape
    k = min(m, n)
    qr, tau, _, info = getattr(_lp, p + "geqrf")(np.asarray(a, order="F"))
    assert info == 0
    q, _, info = getattr(_lp, p + "orgqr")(qr[:, :k], tau)
    assert info == 0
    return np.ascontiguousarray(q)


def lapack_eigh(a: np.ndarray):
    """linalg::eigh -> heev('V','L')  (linalg.rs:39-60).  Returns (eigenvalues ascending, v) where
    ``v`` is the LAPACK output buffer *read back as a row-major array, exactly as the crate does*
    (linalg.rs:54-59): LAPACK leaves eigenvectors in the columns of a column-major buffer, the
    crate reinterprets that buffer row-major, so ``v == Z.T`` (row i of v is eigenvector i)."""
    f = getattr(_lp, _prefix(a.dtype) + "syev")
    w, z, info = f(np.asarray(a, order="F"), compute_v=1, lower=1)
    if info != 0:
        raise LinalgError("cannot compute eigenvalues")
    return w, np.ascontiguousarray(z.T)


def lu_pl(a: np.ndarray):
    """lair::decomposition::lu::Factorized::from(a).into_pl()  (call sites pca.rs:709, 712).
    lair 0.8 is NOT vendored under /root/reference; its documented algorithm is LU with partial
    (row) pivoting, and into_pl() returns P.L with L unit-lower-trapezoidal m x min(m,n).
    Restated with LAPACK getrf.  "parity unpinned" for the factor itself (see module docstring)."""
    m, n = a.shape
    k = min(m, n)
    lu, piv, info = getattr(_lp, _prefix(a.dtype) + "getrf")(np.asarray(a, order="F"))
    if info < 0:
        raise LinalgError("getrf illegal argument")
    l = np.tril(lu[:, :k], -1)
    l[np.arange(k), np.arange(k)] = 1
    # apply the row interchanges in reverse to get P.L
    for i in range(len(piv) - 1, -1, -1):
        j = piv[i]
        if j != i:
            l[[i, j], :] = l[[j, i], :]
    return np.ascontiguousarray(l)


# --------------------------------------------------------------------------- pca.rs
def svd_flip(u: np.ndarray, vt: np.ndarray) -> None:
    """svd_flip (pca.rs:815-850): per column of u, the sign of the FIRST element of maximal
    magnitude (strict '>' update, pca.rs:830) decides; flips u[:, j] and vt[j, :] in place."""
    for j in range(min(u.shape[1], vt.shape[0])):
        col = u[:, j]
        if col.shape[0] == 0:
            continue
        i = int(np.argmax(np.abs(col)))  # numpy argmax returns the first maximum
        if np.sign(col[i]) < 0:
            u[:, j] *= -1
            vt[j, :] *= -1


def randomized_range_finder(x: np.ndarray, omega: np.ndarray, n_iter: int = 7) -> np.ndarray:
    """randomized_range_finder (pca.rs:689-718).  ``omega`` is the d x size StandardNormal draw
    the crate makes at pca.rs:701-705 (row-major fill order), passed in explicitly."""
    q = x @ omega                                   # pca.rs:707
    for _ in range(n_iter):                         # pca.rs:708-715
        q = lu_pl(q)
        pl = q[:, : min(q.shape)]
        q = x.T @ pl
        q = lu_pl(q)
        pl = q[:, : min(q.shape)]
        q = x @ pl
    return lapack_qr_thin(q)                        # pca.rs:716


def randomized_svd(x: np.ndarray, omega: np.ndarray, n_iter: int = 7):
    """randomized_svd (pca.rs:668-686)."""
    q = randomized_range_finder(x, omega, n_iter)
    b = q.T @ x                                     # pca.rs:681
    u, s, vt = lapack_svddc(np.ascontiguousarray(b))  # pca.rs:682
    u = q @ u                                       # pca.rs:683
    svd_flip(u, vt)                                 # pca.rs:684
    return u, s, vt


def transform(x, components, means, centering=True):
    """free fn transform (pca.rs:726-750)."""
    if x.shape[1] != means.shape[0]:
        raise InvalidInput(f"# of columns should be {means.shape[0]}")
    return (x - means) @ components.T if centering else x @ components.T


def transform_with_u(u, singular, n_components):
    """transform_with_u (pca.rs:758-779): U[:, :k] * sigma."""
    return u[:, :n_components] * singular[:n_components]


def inverse_transform(y, components, means, centering=True):
    """inverse_transform (pca.rs:788-811)."""
    if y.shape[1] != components.shape[0]:
        raise InvalidInput(f"# of columns should be {components.shape[0]}")
    return y @ components + means if centering else y @ components


class _PcaBase:
    def __init__(self, n_components: int, centering: bool = True):
        self.n_components = n_components
        self.centering = centering
        self.components = np.zeros((n_components, 0))
        self.means = np.zeros(0)
        self.singular = np.zeros(0)
        self.total_variance = 0.0
        self.n_samples = 0

    def explained_variance_ratio(self):
        """pca.rs:101-105 / 415-419."""
        return self.singular * self.singular / self.total_variance

    def transform(self, x):
        return transform(np.asarray(x), self.components, self.means, self.centering)

    def inverse_transform(self, y):
        return inverse_transform(np.asarray(y), self.components, self.means, self.centering)

    def fit(self, x, **kw):
        self._inner_fit(np.asarray(x), **kw)
        return self

    def fit_transform(self, x, **kw):
        x = np.asarray(x)
        u = self._inner_fit(x, **kw)
        if u.shape[0] == 0 and x.shape[0] == 0:
            return np.zeros((0, 0), dtype=x.dtype) if self.n_components == 0 else u[:, : self.n_components]
        return transform_with_u(u, self.singular, self.n_components)

    def _check(self, x):
        if any(v < self.n_components for v in x.shape):       # pca.rs:199-204 / 513-518
            raise InvalidInput(f"every dimension should be at least {self.n_components}")


class PcaOracle(_PcaBase):
    """Pca<A> (pca.rs:41-232): centre -> gesvd('A','A') -> svd_flip -> top-k.

    thin=True takes the same decomposition from the economy driver (gesdd 'S'): only the first min(n, d) columns of U are
    ever read (svd_flip zips U columns with V^T rows; transform_with_u takes k of them), so every output is the same -- but
    the O(n^2) full U of the crate is not formed, which keeps the oracle to seconds at d = 2048.
    tests/test_oracle_golden.py checks thin against the literal path."""

    def __init__(self, n_components, centering=True, thin=False):
        super().__init__(n_components, centering)
        self.thin = thin

    def _inner_fit(self, x):
        self._check(x)
        n, d = x.shape
        if self.centering:
            if n == 0:                                         # mean_axis -> None (pca.rs:207-211)
                return np.zeros((0, d), dtype=x.dtype)
            means = x.mean(axis=0)
            xc = x - means
        else:
            means = np.zeros(d, dtype=x.dtype)
            xc = x.copy()
        if self.thin:
            u, sigma, vt = lapack_svddc(np.ascontiguousarray(xc))
        else:
            u, sigma, vt = lapack_svd_full(np.ascontiguousarray(xc))   # pca.rs:216-220
        svd_flip(u, vt)                                        # pca.rs:223
        self.total_variance = float(sigma @ sigma)             # pca.rs:224
        self.components = vt[: self.n_components].copy()
        self.n_samples = n
        self.means = means
        self.singular = sigma[: self.n_components].copy()
        return u


class RandomizedPcaOracle(_PcaBase):
    """RandomizedPca<A,R> (pca.rs:317-551).  The Omega draw (pca.rs:701-705) is an argument."""

    def __init__(self, n_components, centering=True, n_oversample=10, n_iter=7):
        super().__init__(n_components, centering)
        self.n_oversample = n_oversample                       # pca.rs:679 hard-codes 10
        self.n_iter = n_iter                                   # pca.rs:680 hard-codes 7

    def _inner_fit(self, x, omega=None, rng=None):
        self._check(x)
        n, d = x.shape
        if self.centering:
            if n == 0:
                return np.zeros((0, d), dtype=x.dtype)
            means = x.mean(axis=0)
            xc = x - means                                     # pca.rs:531
        else:
            means = np.zeros(d, dtype=x.dtype)
            xc = x
        size = self.n_components + self.n_oversample
        if omega is None:
            rng = rng or np.random.default_rng()
            omega = rng.standard_normal((d, size))
        omega = np.asarray(omega, dtype=x.dtype)
        assert omega.shape == (d, size), (omega.shape, (d, size))
        u, sigma, vt = randomized_svd(xc, omega, self.n_iter)  # pca.rs:532
        self.total_variance = float(np.sum(xc.astype(x.dtype) ** 2, dtype=x.dtype))  # pca.rs:533
        self.components = vt[: self.n_components].copy()
        self.n_samples = n
        self.means = means
        self.singular = sigma[: self.n_components].copy()
        return u


# --------------------------------------------------------------------------- ica.rs
def logcosh(wx: np.ndarray):
    """logcosh (ica.rs:383-398): g = tanh(wx) in place; g'_i = mean_j(1 - g_ij^2)."""
    g = np.tanh(wx)
    gp = (1.0 - g * g).sum(axis=1) / wx.shape[1]
    return g, gp


def _normalise_eigenvector_signs(v: np.ndarray) -> np.ndarray:
    """rows of v (= eigenvectors, lapack_eigh's convention) with their first largest-magnitude component made positive"""
    idx = np.argmax(np.abs(v), axis=1)
    sg = np.where(v[np.arange(v.shape[0]), idx] < 0, -1.0, 1.0)
    return v * sg[:, None]


def symmetric_decorrelation(w: np.ndarray, literal: bool = False, normalise_signs: bool = False) -> np.ndarray:
    """symmetric_decorrelation (ica.rs:363-381).

    literal=True reproduces the crate's arithmetic exactly: ``v`` is the heev buffer read
    row-major (= Z^T, see lapack_eigh), its columns are scaled by 1/sqrt(e) and the result is
    ``v . v_saved^T . w = Z^T D Z w``.  literal=False is the textbook
    (W W^T)^(-1/2) W = Z D Z^T W the crate documents.  Identical whenever Z is symmetric (all
    2x2 reference tests)."""
    e, v = lapack_eigh(w @ w.T)
    if normalise_signs:
        # The literal form changes under eigenvector sign flips; LAPACK's raw signs are backend artefacts (MKL in the
        # crate's CI, OpenBLAS here).  normalise_signs=True fixes them the way the device's literal mode does, so the
        # two can be compared element by element for nc > 2 (the textbook form is sign-invariant).
        v = _normalise_eigenvector_signs(v)
    s = 1.0 / np.sqrt(e)
    if literal:
        v_t = v.T.copy()
        return (v * s[None, :]) @ v_t @ w
    z = v.T                                      # columns = eigenvectors
    return (z * s[None, :]) @ z.T @ w


def ica_par(x1: np.ndarray, tol: float, max_iter: int, w_init: np.ndarray, literal: bool = False,
            normalise_signs: bool = False):
    """ica_par (ica.rs:319-361).  x1 is nc x n (whitened).  Returns (W, n_iter).

    literal=True uses the crate's convergence test rows(W1) . columns(W) (ica.rs:345-349, Q4)
    and the literal decorrelation; literal=False the textbook rows . rows."""
    w = symmetric_decorrelation(w_init, literal, normalise_signs)
    p_inv = 1.0 / x1.shape[1]
    for i in range(max_iter):
        g, gp = logcosh(w @ x1)
        d = g @ x1.T * p_inv - gp[:, None] * w
        w1 = symmetric_decorrelation(d, literal, normalise_signs)
        if literal:
            dots = np.einsum("ij,ji->i", w1, w)
        else:
            dots = np.einsum("ij,ij->i", w1, w)
        lim = np.max(np.abs(np.abs(dots) - 1.0))
        if lim < tol:
            return w1, i + 1
        w = w1
    return w, max_iter


class FastIcaOracle:
    """FastIca<A,R> (ica.rs:41-221).  w_init (ica.rs:210-214) is an argument; ``n_components``
    (absent in the crate, which always uses min(n,d), ica.rs:173) keeps the first rows of K."""

    def __init__(self, n_components=None, tol=1e-4, max_iter=200, literal=False, whiten="svd"):
        self.n_components = n_components
        self.tol, self.max_iter, self.literal, self.whiten = tol, max_iter, literal, whiten
        self.components = np.zeros((0, 0))
        self.means = np.zeros(0)
        self.n_iter = 0

    def whitening(self, x):
        n, d = x.shape
        nc = min(n, d) if self.n_components is None else self.n_components
        means = x.mean(axis=0)
        xt = np.ascontiguousarray((x - means).T)               # ica.rs:178-188, d x n
        if self.whiten == "svd":
            u, sigma = lapack_svd_left(xt)                     # ica.rs:189
        else:  # mathematically identical, O(n d^2) -> O(d^3): eigh of the d x d covariance
            lam, v = np.linalg.eigh(xt @ xt.T)
            order = np.argsort(lam)[::-1]
            u, sigma = v[:, order], np.sqrt(np.maximum(lam[order], 0))
        k = (u[:, :nc] / sigma[:nc]).T                         # ica.rs:190-203
        x1 = k @ xt * np.sqrt(n)                               # ica.rs:204-208
        return means, xt, k, x1

    def fit(self, x, w_init=None, rng=None):
        self._inner_fit(np.asarray(x), w_init, rng)
        return self

    def _inner_fit(self, x, w_init=None, rng=None):
        n, d = x.shape
        if n == 0:                                             # ica.rs:174-176
            return np.zeros((0, d), dtype=x.dtype)
        means, xt, k, x1 = self.whitening(x)
        nc = k.shape[0]
        if w_init is None:
            rng = rng or np.random.default_rng()
            w_init = rng.standard_normal((nc, nc))
        w_init = np.asarray(w_init, dtype=x.dtype)
        w, n_iter = ica_par(x1, self.tol, self.max_iter, w_init, self.literal)  # ica.rs:216
        self.components = w @ k                                # ica.rs:217
        self.means = means
        self.n_iter = n_iter
        self.k_, self.w_, self.x1_ = k, w, x1
        return xt

    def transform(self, x):
        x = np.asarray(x)
        if x.shape[1] != self.means.shape[0]:
            raise InvalidInput("too many columns")             # ica.rs:124-128
        return (x - self.means) @ self.components.T            # ica.rs:129-130

    def fit_transform(self, x, w_init=None, rng=None):
        xt = self._inner_fit(np.asarray(x), w_init, rng)
        return (self.components @ xt).T.copy()                 # ica.rs:155-156


# --------------------------------------------------------------------------- synthetic inputs
try:  # the seeded generators live in the neutral synth_data module at the repo root
    from synth_data import synth_ica, synth_pca  # noqa: F401
except ImportError:  # pragma: no cover
    pass

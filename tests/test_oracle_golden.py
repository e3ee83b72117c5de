"""Pins oracle/petal_oracle.py against every known-answer test the reference holds for the hot
path (tests/golden/reference_kats.json).  CPU only."""
import numpy as np
import pytest

from oracle import petal_oracle as po


def test_pca_zero_component(kats):
    # src/pca.rs:862-875
    pca = po.PcaOracle(0)
    y = pca.fit_transform(np.zeros((0, 5), dtype=np.float32))
    assert y.shape == (0, 0)
    y = pca.fit_transform(np.array(kats["pca_zero_component"]["cases"][1]["x"], dtype=np.float32))
    assert y.shape == (3, 0)


def test_pca_single_sample(kats):
    c = kats["pca_single_sample"]
    y = po.PcaOracle(1).fit_transform(np.array(c["x"], dtype=np.float32))
    assert np.array_equal(y, np.array(c["y"], dtype=np.float32))


def test_pca(kats):
    c = kats["pca"]
    x = np.array(c["x"], dtype=np.float64)
    pca = po.PcaOracle(1)
    y = pca.fit_transform(x)
    assert np.allclose(np.abs(y[:, 0]), c["abs_y"], atol=c["tol"], rtol=0)
    assert np.allclose(pca.inverse_transform(y), x, atol=c["tol"], rtol=0)
    pca = po.PcaOracle(1).fit(x)
    comp = pca.components
    ref = np.array(c["components"])
    assert min(np.abs(comp - ref).max(), np.abs(comp + ref).max()) < c["tol"]   # sign: SURVEY Q5
    y = pca.transform(x)
    assert np.allclose(np.abs(y[:, 0]), c["abs_y"], atol=c["tol"], rtol=0)


def test_pca_without_centering(kats):
    c = kats["pca_without_centering"]
    y = po.PcaOracle(1, centering=False).fit_transform(np.array(c["x"], dtype=np.float64))
    assert np.allclose(np.abs(y[:, 0]), c["abs_y"], atol=c["tol"], rtol=0)


@pytest.mark.parametrize("cls", ["exact", "randomized"])
def test_explained_variance_ratio(kats, cls):
    c = kats["explained_variance_ratio"]
    x = np.array(c["x"], dtype=np.float64)
    if cls == "exact":
        m = po.PcaOracle(2).fit(x)
    else:
        m = po.RandomizedPcaOracle(2).fit(x, rng=np.random.default_rng(0))
    r = m.explained_variance_ratio()
    assert r[0] > c["ratio0_gt"] and r[1] < c["ratio1_lt"]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_randomized_pca(kats, seed):
    c = kats["randomized_pca"]
    x = np.array(c["x"], dtype=np.float64)
    pca = po.RandomizedPcaOracle(1).fit(x, rng=np.random.default_rng(seed))
    y = pca.transform(x)
    assert np.allclose(np.abs(y[:, 0]), c["abs_y"], atol=c["tol"], rtol=0)
    assert np.allclose(pca.inverse_transform(y), x, atol=c["tol"], rtol=0)
    y = po.RandomizedPcaOracle(1).fit_transform(x, rng=np.random.default_rng(seed + 10))
    assert np.allclose(np.abs(y[:, 0]), c["abs_y"], atol=c["tol"], rtol=0)


def test_randomized_vs_exact(kats):
    c = kats["randomized_vs_exact"]
    rng = np.random.default_rng(1234567891011121314 % 2**32)
    x = rng.standard_normal(tuple(c["shape"]))
    a = po.PcaOracle(c["k"]).fit(x)
    b = po.RandomizedPcaOracle(c["k"]).fit(x, rng=rng)
    assert np.allclose(a.explained_variance_ratio(), b.explained_variance_ratio(), rtol=c["max_relative"])
    assert np.allclose(a.singular, b.singular, rtol=c["max_relative"])


def test_svd_flip(kats):
    c = kats["svd_flip"]
    u, v = np.array(c["u"], dtype=float), np.array(c["v"], dtype=float)
    po.svd_flip(u, v)
    assert np.array_equal(u, np.array(c["u_out"], dtype=float))
    assert np.array_equal(v, np.array(c["v_out"], dtype=float))


@pytest.mark.parametrize("literal", [False, True])
def test_ica_par_single_iter(kats, literal):
    c = kats["ica_par_single_iter"]
    w, n = po.ica_par(np.array(c["x"]), c["tol"], c["max_iter"], np.array(c["w_init"], dtype=float), literal)
    assert n == c["n_iter"]
    assert np.allclose(w, c["w"], atol=c["abs_tol"], rtol=0)


@pytest.mark.parametrize("literal", [False, True])
def test_ica_par_multi_iter(kats, literal):
    c = kats["ica_par_multi_iter"]
    w, n = po.ica_par(np.array(c["x"], dtype=float), c["tol"], c["max_iter"], np.array(c["w_init"], dtype=float), literal)
    assert n == c["n_iter"]
    assert np.allclose(w, c["w"], atol=c["abs_tol"], rtol=0)


def test_logcosh(kats):
    c = kats["logcosh"]
    g, gp = po.logcosh(np.array(c["x"], dtype=float))
    assert np.allclose(g, c["g"], rtol=c["g_rel"], atol=0)
    assert np.allclose(gp, c["gp"], rtol=c["gp_rel"], atol=0)


@pytest.mark.parametrize("literal", [False, True])
def test_symmetric_decorrelation(kats, literal):
    c = kats["symmetric_decorrelation"]
    w = po.symmetric_decorrelation(np.array(c["x"], dtype=float), literal)
    assert np.allclose(w, c["w"], rtol=c["rel"], atol=0)


def test_fast_ica_fit_transform(kats):
    # src/ica.rs:407-420: fit()+transform() == fit_transform() for the same w_init
    c = kats["fast_ica_fit_transform"]
    x = np.array(c["x"], dtype=float)
    w0 = np.random.default_rng(3).standard_normal((2, 2))
    a = po.FastIcaOracle().fit(x, w_init=w0)
    b = po.FastIcaOracle()
    yb = b.fit_transform(x, w_init=w0)
    assert a.n_iter == b.n_iter
    assert np.allclose(a.transform(x), yb, atol=1e-12)


def test_whitening_paths_agree():
    x = po.synth_ica(400, 6, 6, seed=5, dtype=np.float64)
    a = po.FastIcaOracle(whiten="svd")
    b = po.FastIcaOracle(whiten="eigh")
    _, _, ka, x1a = a.whitening(x)
    _, _, kb, x1b = b.whitening(x)
    s = np.sign(np.sum(ka * kb, axis=1))
    assert np.allclose(ka, kb * s[:, None], atol=1e-8)
    assert np.allclose(np.cov(x1a, bias=True), np.eye(6), atol=1e-8)


def test_oracle_fp32_vs_fp64_same_omega():
    # the 1e-5 parity target is reachable in fp32
    x32 = po.synth_pca(4000, 96, 8, seed=11, dtype=np.float32)
    om = np.random.default_rng(3).standard_normal((96, 18))
    a = po.RandomizedPcaOracle(8, n_iter=5).fit(x32.astype(np.float64), omega=om)
    b = po.RandomizedPcaOracle(8, n_iter=5).fit(x32, omega=om.astype(np.float32))
    rel = np.linalg.norm(a.components - b.components, axis=1) / np.linalg.norm(a.components, axis=1)
    assert rel.max() < 1e-5
    assert np.allclose(a.singular, b.singular, rtol=1e-5)


def test_thin_pca_oracle_equals_the_literal_full_svd_path():
    """PcaOracle(thin=True) (gesdd 'S', used where the crate's full n x n U would take minutes) gives the outputs of the
    literal gesvd('A','A') path: components, singular values, ratios, fit_transform -- signs included."""
    x = po.synth_pca(300, 40, 5, seed=9, dtype=np.float64)
    a, b = po.PcaOracle(5), po.PcaOracle(5, thin=True)
    ya, yb = a.fit_transform(x), b.fit_transform(x)
    assert np.allclose(a.components, b.components, atol=1e-10)
    assert np.allclose(a.singular, b.singular, rtol=1e-12)
    assert np.allclose(a.explained_variance_ratio(), b.explained_variance_ratio(), rtol=1e-10)
    assert np.allclose(ya, yb, atol=1e-9 * np.abs(ya).max())

"""CPU suite for the host side: the product's algo.cpp/api.cpp run against the host-memory simulation of
the device-op layer (oracle/cpu_ops.cpp, test infrastructure).  Checks the reference's known-answer
tests and the oracle parity of the host algorithms; the HIP kernels themselves are covered by -m gpu."""
import numpy as np
import pytest

import hostsim
import parity_cases as pc


@pytest.fixture(scope="module")
def ctx():
    return hostsim.context()


@pytest.mark.parametrize("case", pc.ALL_KATS, ids=lambda f: f.__name__)
def test_reference_kats(ctx, kats, case):
    case(ctx, kats)


def test_gemm_ops(ctx):
    pc.gemm_exact(ctx, 37, 24, 7)
    pc.gemm_exact(ctx, 130, 48, 33)


@pytest.mark.parametrize("n,d,k,n_iter", [(2000, 64, 8, 5), (1501, 50, 5, 7), (300, 40, 12, 3)])
def test_rpca_parity(ctx, n, d, k, n_iter):
    pc.rpca_parity(ctx, n, d, k, n_iter, seed=n)


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
@pytest.mark.parametrize("spectrum", ["planted", "geo97", "rsqrt"])
@pytest.mark.parametrize("n_iter", [0, 1, 2])
def test_rpca_low_iteration_counts(ctx, n_iter, spectrum, mode):
    """n_iter 0 (the Cholesky-QR2 branch of the raw sketch), 1 and 2 against the oracle with the same Omega, with the
    split-product mode's roundings of the sketch matrix and of the re-based iterate simulated (oracle/cpu_ops.cpp)"""
    ctx.set_gemm_mode(mode)
    try:
        pc.rpca_low_iter(ctx, 1500, 96, 12, n_iter, spectrum, np.float32, seed=11 + n_iter)
        pc.rpca_low_iter(ctx, 900, 64, 8, n_iter, spectrum, np.float64, seed=21 + n_iter)
    finally:
        ctx.set_gemm_mode("fp32")


def test_two_plane_verdict_and_exact_redo(ctx):
    pc.two_plane_verdict_case(ctx)


def test_close_eigenvalues_repeat_the_small_stage_only(ctx):
    """The eigen-solver's closeness verdict (forced by the simulation's hook, PETAL_OPT_EIGH_JACOBI = 2; on the device:
    tests/test_gpu_parity.py::test_close_eigenvalues_repeat_the_small_eigen_solve_only) repeats the small stage of a RandomizedPca fit --
    eigen-solve, verdicts, components, U, svd_flip -- with the Jacobi solver; the passes over X stand (rpca_redo = 0).  Also when the
    heavy-tail verdict sends the same fit through the exact pipeline first."""
    pc.rpca_parity(ctx, 1200, 64, 8, 4, seed=31)
    base = (pc.rpca_parity.last_fit_stats["eigh_redo"], pc.rpca_parity.last_fit_stats["rpca_redo"])
    assert base == (0, 0), base
    ctx.set_option("eigh_jacobi", 2)
    try:
        pc.rpca_parity(ctx, 1200, 64, 8, 4, seed=31)
        st = pc.rpca_parity.last_fit_stats
        assert st["eigh_redo"] == 1 and st["rpca_redo"] == 0, st
        ctx.set_gemm_mode("bf16x3")
        ctx.set_option("verdict_threshold", 1e-12)      # (every optimistic run is "heavy-tailed": exact redo, then the small stage again)
        pc.rpca_parity(ctx, 1200, 64, 8, 4, seed=31)
        st = pc.rpca_parity.last_fit_stats
        assert st["eigh_redo"] == 1 and st["rpca_redo"] == 1, st
    finally:
        ctx.set_option("eigh_jacobi", 0)
        ctx.set_option("verdict_threshold", 4e-6)
        ctx.set_gemm_mode("fp32")


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_exact_pca_keeps_small_singular_values_of_uncentred_data(ctx, dtype):
    pc.pca_wide_uncentred_case(ctx, dtype)


@pytest.mark.parametrize("n,d,nc,dtype,offset", [(2000, 8, 8, np.float64, 0.0), (5000, 40, 12, np.float64, 40.0), (3000, 100, 5, np.float32, 0.0),
                                               (2000, 130, 24, np.float32, 3.0)])
def test_fastica_on_the_oracles_trajectory(ctx, n, d, nc, dtype, offset):
    pc.ica_strict_parity(ctx, n, d, nc, seed=600 + nc, dtype=dtype, offset=offset)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_inputs_without_a_factorisation(ctx, dtype):
    pc.degenerate_input_case(ctx, 500, 40, dtype)


@pytest.mark.parametrize("mode", ["bf16x3", "fp32"])
def test_lost_pivot_retries_with_the_sketch_rebased(ctx, mode):
    ctx.set_gemm_mode(mode)
    try:
        pc.rebased_retry_case(ctx, expect_retry=True)
    finally:
        ctx.set_gemm_mode("fp32")


def test_components_beyond_a_ring_slot(ctx):
    """k d esz above the 8 MiB ring slot (the single result view threw there): the components leave by their own
    copy.  k = 512, d = 2048 fp64 = 8 MiB + the small block."""
    pc.rpca_parity(ctx, 700, 2048, 512, 1, seed=5, dtype=np.float64, tol=1e-7)


@pytest.mark.parametrize("layout", ["fortran", "strided"])
def test_host_inputs_that_are_not_row_major(ctx, layout):
    pc.rpca_parity(ctx, 900, 40, 6, 4, seed=9, host_layout=layout)
    pc.rpca_parity(ctx, 500, 24, 4, 4, seed=10, dtype=np.float64, tol=1e-9, host_layout=layout)


def test_means_folded_into_the_first_fused_pass(ctx):
    ctx.set_option("means_fold_rows", 0)       # (the product folds from 200000 rows on)
    ctx.set_gemm_mode("bf16x3")
    try:
        pc.means_fold_case(ctx, 2500, 64, 5, expect_folded=True)   # (l = 15: the fold needs a spare padding column, l < LP)
        pc.rpca_parity(ctx, 2000, 48, 6, 5, seed=83)
        ctx.set_option("means_fold_rows", -1)
        pc.means_fold_case(ctx, 2500, 64, 5, expect_folded=False)
    finally:
        ctx.set_gemm_mode("fp32")
        ctx.set_option("means_fold_rows", 200000)


def test_fastica_whitening_from_the_split_product_covariance(ctx):
    ctx.set_gemm_mode("bf16x3")
    try:
        pc.ica_split_gram_case(ctx, 3000, 384, 5)
    finally:
        ctx.set_gemm_mode("fp32")


def test_steering_passes_on_sixteen_bit_operands(ctx, monkeypatch):
    """the host sequencing of the steering passes (algo.cpp marks every fused pass but the last one) on the simulation's rounding"""
    ctx.set_gemm_mode("bf16x3")
    try:
        assert pc.steering_pass_case(ctx, monkeypatch, 1500, 64, 6, 5, seed=95, tol=2e-5) == 0
        assert pc.steering_pass_case(ctx, monkeypatch, 1200, 128, 90, 4, seed=98, tol=5e-5) == 0   # (l = 100: the K1 / K2 steering products)
    finally:
        ctx.set_gemm_mode("fp32")


def test_fastica_means_gathered_in_the_gram_pass(ctx):
    """the host logic of the folded means (algo.cpp: fastica_fit) on the simulation's restatement of the device arithmetic (the case
    asserts petal_stats.means_folded = 1), then the SAME fit with the fold switched off (PETAL_OPT_MEANS_FOLD_ROWS < 0: a means pass of
    its own, means_folded = 0): same sources"""
    from parity_cases import petal, np, po
    ctx.set_gemm_mode("bf16x3")
    try:
        y = pc.ica_means_fold_case(ctx, 3000, 272, 5)
        ctx.set_option("means_fold_rows", -1)
        x = po.synth_ica(3000, 272, 5, seed=71, dtype=np.float64)
        x = (x + 40.0 * x.std(axis=0) * np.sign(np.random.default_rng(72).standard_normal(272))).astype(np.float32)
        w0 = np.random.default_rng(73).standard_normal((5, 5)).astype(np.float32)
        y2 = np.asarray(petal.FastIca(ctx=ctx, n_components=5).fit_transform(x, w_init=w0))
        assert ctx.stats()["means_folded"] == 0, ctx.stats()
        c = np.abs(np.corrcoef(y.T.astype(np.float64), y2.T.astype(np.float64))[:5, 5:])
        assert np.abs(1.0 - c.max(axis=1)).max() <= 1e-3, c.max(axis=1)
    finally:
        ctx.set_gemm_mode("fp32")
        ctx.set_option("means_fold_rows", 200000)
    assert y.shape == (3000, 5)


def test_power_pass_entry(ctx):
    """petal_power_pass through the host simulation: the fused form (split-product mode) and the K1 + K2 fall-back"""
    assert pc.power_pass_exact(ctx, 300, 48, 20, seed=1) is False
    ctx.set_gemm_mode("bf16x3")
    try:
        assert pc.power_pass_exact(ctx, 300, 48, 20, seed=2) is True
        assert pc.power_pass_exact(ctx, 300, 48, 100, seed=3) is False
    finally:
        ctx.set_gemm_mode("fp32")


def test_rpca_f64_and_no_centering(ctx):
    pc.rpca_parity(ctx, 800, 32, 4, 7, seed=3, dtype=np.float64, tol=1e-9)
    pc.rpca_parity(ctx, 800, 32, 4, 7, seed=4, centering=False)


def test_pca_parity(ctx):
    pc.pca_parity(ctx, 1000, 16, 4, seed=1)                       # BASELINE configs[0]
    pc.pca_parity(ctx, 500, 24, 6, seed=2, dtype=np.float32, tol=2e-5)


def test_ica_parity(ctx):
    pc.ica_parity(ctx, 3000, 6, 6, seed=5, dtype=np.float64)
    pc.ica_parity(ctx, 3000, 12, 4, seed=6, dtype=np.float32, n_components=4)
    pc.ica_par_parity(ctx, 2000, 5, seed=8, dtype=np.float64, tol=1e-8)


def test_topk_subspace_eigensolver_paths(ctx):
    # d > 88 and k / n_components << d: whitening and exact Pca go through the block subspace iteration
    pc.pca_parity(ctx, 1500, 128, 6, seed=31, dtype=np.float64, tol=1e-8)
    pc.ica_parity(ctx, 3000, 100, 6, seed=32, dtype=np.float64, n_components=6)


def test_edge_cases(ctx):
    pc.edge_cases(ctx)


@pytest.mark.parametrize("nc", [3, 4, 8])
def test_ica_literal_mode_matches_the_literal_oracle(ctx, nc):
    pc.ica_literal_parity(ctx, nc, seed=nc)


def test_ica_literal_convergence_test_at_nc2(ctx):
    pc.ica_literal_convergence_nc2(ctx)


def test_accurate_route_for_ill_conditioned_fp64(ctx):
    pc.accurate_route_case(ctx)


def test_wide_spectrum_robust_path_refills_dropped_directions(ctx):
    """sigma_1 / sigma_l = 10^5: the optimistic single-Cholesky re-basing breaks down (cond(Yp^T Yp) beyond fp64) and the robust
    redo's dependence test drops the tail of the block at the first iteration (every column of Xc Omega is dominated by
    sigma_1); the dropped columns are refilled with fresh directions, so the block regains its width and ALL k components come
    back (round 2 returned sigma = 0 for the smallest ones) -- the crate's pivoted LU (src/pca.rs:709-713) never loses rank."""
    import petal_decomposition_amd as petal
    from oracle import petal_oracle as po
    rng = np.random.default_rng(91)
    n, d, k = 4000, 128, 16
    u, _ = np.linalg.qr(rng.standard_normal((n, 2 * k)))
    v, _ = np.linalg.qr(rng.standard_normal((d, 2 * k)))
    s = 10.0 ** (-4.0 * np.arange(2 * k) / (k - 1))
    x = ((u * s) @ v.T * 100.0).astype(np.float32)
    om = rng.standard_normal((d, k + 10))
    o = po.RandomizedPcaOracle(k, centering=False, n_iter=7).fit(x.astype(np.float64), omega=om)
    m = petal.RandomizedPca(k, centering=False, ctx=ctx, n_iter=7).fit(x, omega=om.astype(np.float32))
    assert np.all(m.singular_values() > 0)
    assert np.abs(m.singular_values() / o.singular - 1.0).max() < 1e-4
    assert pc.rowwise_rel(m.components().astype(np.float64), o.components).max() < 1e-3

"""Workloads of the multi-rank HIP test: shared by the rank workers (tests/sharded_gpu_worker.py, which see a row
block) and by the parent test (tests/test_gpu_sharded.py, which sees the whole matrix, the single-process HIP fit and
the oracle).  Seeds are fixed, so every process regenerates the same matrices."""
import numpy as np

from synth_data import synth_ica, synth_pca

RPCA = dict(n=6001, d=64, k=6, n_iter=4)
OUTLIERS = (5207, 1033)   # one row in rank 1's block and one in rank 0's (2 ranks: the blocks meet at row 3803)


def x_rpca32():
    """planted model + two outlier rows, so that the first max-|.| element svd_flip looks for (pca.rs:826-839) sits in
    rank 1's row block for some columns of U and in rank 0's for others"""
    x = synth_pca(RPCA["n"], RPCA["d"], RPCA["k"], seed=77, dtype=np.float64)
    mu = x.mean(0)
    v = np.linalg.svd(x - mu, full_matrices=False)[2]
    x[OUTLIERS[0]] = mu + 9.0 * (x[OUTLIERS[0]] - mu) + 2.0 * np.abs(x - mu).max() * (v[0] + v[2] - v[4])
    x[OUTLIERS[1]] = mu + 9.0 * (x[OUTLIERS[1]] - mu) + 2.0 * np.abs(x - mu).max() * (v[1] - v[3] + v[5])
    return x.astype(np.float32)


def omega_rpca(dtype):
    return np.random.default_rng(5).standard_normal((RPCA["d"], RPCA["k"] + 10)).astype(dtype)


def run_rpca(dtype, explicit):
    def run(petal, ctx, xs, rank):
        if explicit:
            m = petal.RandomizedPca(RPCA["k"], ctx=ctx, n_iter=RPCA["n_iter"])
            y = m.fit_transform(xs, omega=omega_rpca(dtype))
        else:   # every rank draws its OWN Omega from a differently seeded generator: rank 0's must win
            m = petal.RandomizedPca(RPCA["k"], ctx=ctx, n_iter=RPCA["n_iter"], rng=np.random.default_rng(100 + rank))
            y = m.fit_transform(xs)
        return {"components": m.components(), "singular": m.singular_values(), "mean": m.mean(),
                "evr": m.explained_variance_ratio(), "y": y}
    return run


def x_pca():
    return synth_pca(2000, 32, 3, seed=78, dtype=np.float64)


def run_pca(petal, ctx, xs, rank):
    p = petal.Pca(3, ctx=ctx)
    y = p.fit_transform(xs)
    return {"components": p.components(), "singular": p.singular_values(), "evr": p.explained_variance_ratio(), "y": y}


ICA = dict(n=30000, d=12, nc=8)


def x_ica():
    return synth_ica(ICA["n"], ICA["d"], ICA["nc"], seed=79, dtype=np.float32)


def w0_ica():
    return np.random.default_rng(9).standard_normal((ICA["nc"], ICA["nc"])).astype(np.float32)


def run_ica(explicit):
    def run(petal, ctx, xs, rank):
        if explicit:
            ica = petal.FastIca(ctx=ctx, n_components=ICA["nc"])
            y = ica.fit_transform(xs, w_init=w0_ica())
        else:
            ica = petal.FastIca(np.random.default_rng(200 + rank), ctx, n_components=ICA["nc"])
            y = ica.fit_transform(xs)
        return {"components": ica.components, "mean": ica.means, "n_iter": np.array([ica.n_iter]), "y": y}
    return run


# ---- the per-rank workloads of BASELINE configs[3] / configs[4], split over the ranks of the test ------------------------
CFG4 = dict(n=250000, d=1024, k=128, n_iter=7)    # one of the eight ranks' rows of configs[3]
CFG5 = dict(n=500000, d=512, nc=64)               # one of the eight ranks' rows of configs[4]


def x_cfg4():
    return synth_pca(CFG4["n"], CFG4["d"], CFG4["k"], seed=4, dtype=np.float32)


def omega_cfg4():
    return np.random.default_rng(3).standard_normal((CFG4["d"], CFG4["k"] + 10)).astype(np.float32)


def run_cfg4(petal, ctx, xs, rank):
    m = petal.RandomizedPca(CFG4["k"], ctx=ctx, n_iter=CFG4["n_iter"])
    m.fit(xs, omega=omega_cfg4())
    return {"components": m.components(), "singular": m.singular_values(), "evr": m.explained_variance_ratio(), "mean": m.mean()}


def x_cfg5():
    return synth_ica(CFG5["n"], CFG5["d"], CFG5["nc"], seed=8, dtype=np.float32)


def w0_cfg5():
    return np.random.default_rng(7).standard_normal((CFG5["nc"], CFG5["nc"])).astype(np.float32)


def run_cfg5(petal, ctx, xs, rank):
    ica = petal.FastIca(ctx=ctx, n_components=CFG5["nc"])
    ica.fit(xs, w_init=w0_cfg5())
    return {"components": ica.components, "mean": ica.means, "n_iter": np.array([ica.n_iter])}


# UNCENTRED data 3 sigma off centre, k = 100 of 512 features, n_iter = 3: the eigen-solver's closeness verdict fires (round 6), every rank
# must repeat the SMALL stage (agreed code 2 on the svd_flip key's all-reduce) and nothing else
EIG = dict(n=12000, d=512, k=100, n_iter=3)


def x_eig():
    x = synth_pca(EIG["n"], EIG["d"], EIG["k"], seed=9500, dtype=np.float64)
    return (x + 3.0 * x.std(axis=0) * np.sign(np.random.default_rng(7).standard_normal(EIG["d"]))).astype(np.float32)


def omega_eig():
    return np.random.default_rng(10500).standard_normal((EIG["d"], EIG["k"] + 10)).astype(np.float32)


def run_eig(petal, ctx, xs, rank):
    m = petal.RandomizedPca(EIG["k"], centering=False, ctx=ctx, n_iter=EIG["n_iter"])
    m.fit(xs, omega=omega_eig())
    st = ctx.stats()
    return {"components": m.components(), "singular": m.singular_values(), "redo": np.array([st["eigh_redo"], st["rpca_redo"]])}


CASES = {
    "rpca32_close_eigenvalues": {"x": x_eig, "run": run_eig},
    "rpca32": {"x": x_rpca32, "run": run_rpca(np.float32, True), "both_modes": True},
    "rpca32_rerun": {"x": x_rpca32, "run": run_rpca(np.float32, True), "both_modes": True},   # run-to-run determinism
    "rpca32_own_omega": {"x": x_rpca32, "run": run_rpca(np.float32, False)},
    "rpca64": {"x": lambda: x_rpca32().astype(np.float64), "run": run_rpca(np.float64, True)},
    "pca64": {"x": x_pca, "run": run_pca},
    "pca32": {"x": lambda: x_pca().astype(np.float32), "run": run_pca},
    "ica32": {"x": x_ica, "run": run_ica(True), "both_modes": True},
    "ica32_rerun": {"x": x_ica, "run": run_ica(True), "both_modes": True},
    "ica32_own_w": {"x": x_ica, "run": run_ica(False)},
    "cfg4_share": {"x": x_cfg4, "run": run_cfg4},
    "cfg5_share": {"x": x_cfg5, "run": run_cfg5},
}

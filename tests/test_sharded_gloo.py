"""Sample-sharded path on CPU: two processes (gloo), each holding a row block of X, run the PRODUCT's host
algorithms over the host-memory simulation of the device ops and all-reduce the small replicated matrices
through the same collective hook the GPU path uses with RCCL.  The sharded result must equal the
single-process result and the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import hostsim
    import petal_decomposition_amd as petal
    from synth_data import synth_ica, synth_pca
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ctx = hostsim.context()
    ctx.use_torch_distributed()
    res = {}
    # RandomizedPca: uneven shards (rank 0 gets more rows)
    n, d, k = 3001, 48, 6
    x = synth_pca(n, d, k, seed=77, dtype=np.float32)
    cut = [0, 1700, n]
    xs = x[cut[rank]:cut[rank + 1]]
    om = np.random.default_rng(5).standard_normal((d, k + 10)).astype(np.float32)
    m = petal.RandomizedPca(k, ctx=ctx, n_iter=4)
    y = m.fit_transform(xs, omega=om)
    res["rpca_components"], res["rpca_singular"], res["rpca_mean"] = m.components(), m.singular_values(), m.mean()
    res["rpca_evr"], res["rpca_y"] = m.explained_variance_ratio(), y
    # the eigen-solver's closeness verdict (forced by the simulation's hook): every rank repeats the SMALL stage with the Jacobi solver --
    # the agreed code rides the svd_flip key's all-reduce -- and nothing else; same numbers as the fit above
    ctx.set_option("eigh_jacobi", 2)
    m3 = petal.RandomizedPca(k, ctx=ctx, n_iter=4)
    y3 = m3.fit_transform(xs, omega=om)
    st3 = ctx.stats()
    ctx.set_option("eigh_jacobi", 0)
    res["eig_components"], res["eig_singular"], res["eig_y"] = m3.components(), m3.singular_values(), y3
    res["eig_stats"] = np.array([st3["eigh_redo"], st3["rpca_redo"]])
    # rank-deficient fp32 data (rank 3, l = 16): a pivot is lost, every rank retries with the sketch re-based (agreed code 3), loses it again
    # and takes the robust path -- three runs, each rank the same collectives
    rl = np.random.default_rng(12)
    xl = (rl.standard_normal((n, 3)) @ rl.standard_normal((3, d))).astype(np.float32)
    m4 = petal.RandomizedPca(k, ctx=ctx, n_iter=4).fit(xl[cut[rank]:cut[rank + 1]], omega=om)
    res["low_components"], res["low_singular"], res["low_redo"] = m4.components(), m4.singular_values(), np.array([ctx.stats()["rpca_redo"]])
    # exact Pca, f64
    x64 = synth_pca(900, 12, 3, seed=78, dtype=np.float64)
    xs64 = x64[rank * 450:(rank + 1) * 450]
    p = petal.Pca(3, ctx=ctx)
    res["pca_y"] = p.fit_transform(xs64)
    res["pca_components"], res["pca_singular"] = p.components(), p.singular_values()
    # FastICA
    xi = synth_ica(4000, 5, 5, seed=79, dtype=np.float64)
    xis = xi[rank * 2000:(rank + 1) * 2000]
    w0 = np.random.default_rng(9).standard_normal((5, 5))
    ica = petal.FastIca(ctx=ctx)
    res["ica_y"] = ica.fit_transform(xis, w_init=w0)
    res["ica_components"], res["ica_n_iter"] = ica.components, np.array([ica.n_iter])
    # no explicit Omega / w_init: every rank's model owns a differently seeded generator; the library must replicate
    # rank 0's draw or the ranks diverge (FastICA: different stop iterations -> a hung all-reduce)
    m2 = petal.RandomizedPca(k, ctx=ctx, n_iter=4, rng=np.random.default_rng(100 + rank)).fit(xs)
    res["own_rpca_components"], res["own_rpca_singular"] = m2.components(), m2.singular_values()
    ica2 = petal.FastIca(np.random.default_rng(200 + rank), ctx).fit(xis)
    res["own_ica_components"], res["own_ica_n_iter"] = ica2.components, np.array([ica2.n_iter])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_matches_single(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostsim
    hostsim.build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")

    import petal_decomposition_amd as petal
    from oracle import petal_oracle as po
    from synth_data import synth_ica, synth_pca
    import parity_cases as pc
    ctx = hostsim.context()
    # every replicated result is identical on both ranks
    for key in ("rpca_components", "rpca_singular", "rpca_mean", "rpca_evr", "pca_components", "pca_singular",
                "ica_components", "ica_n_iter", "own_rpca_components", "own_rpca_singular", "own_ica_components",
                "own_ica_n_iter"):
        assert np.array_equal(r0[key], r1[key]), key

    n, d, k = 3001, 48, 6
    x = synth_pca(n, d, k, seed=77, dtype=np.float32)
    om = np.random.default_rng(5).standard_normal((d, k + 10)).astype(np.float32)
    single = petal.RandomizedPca(k, ctx=ctx, n_iter=4)
    ys = single.fit_transform(x, omega=om)
    assert pc.rowwise_rel(r0["rpca_components"], single.components()).max() < 2e-6
    assert np.allclose(r0["rpca_singular"], single.singular_values(), rtol=2e-6)
    assert np.allclose(r0["rpca_evr"], single.explained_variance_ratio(), rtol=1e-5)
    assert np.allclose(r0["rpca_mean"], single.mean(), atol=1e-6)
    y = np.concatenate([r0["rpca_y"], r1["rpca_y"]])
    assert np.abs(y - ys).max() <= 1e-4 * np.abs(ys).max()      # includes the cross-rank svd_flip decision
    o = po.RandomizedPcaOracle(k, n_iter=4).fit(x.astype(np.float64), omega=om.astype(np.float64))
    assert pc.rowwise_rel(r0["rpca_components"].astype(np.float64), o.components).max() < 1e-5

    for r in (r0, r1):   # the repeated small stage: flagged on both ranks, no redo of the passes, the first fit's numbers
        assert r["eig_stats"].tolist() == [1, 0], r["eig_stats"]
        assert np.array_equal(r["eig_components"], r["rpca_components"]) and np.array_equal(r["eig_singular"], r["rpca_singular"])
        assert np.array_equal(r["eig_y"], r["rpca_y"])

    rl = np.random.default_rng(12)
    xl = (rl.standard_normal((n, 3)) @ rl.standard_normal((3, d))).astype(np.float32)
    low = petal.RandomizedPca(k, ctx=ctx, n_iter=4).fit(xl, omega=om)
    assert ctx.stats()["rpca_redo"] == 2 and int(r0["low_redo"][0]) == 2 and int(r1["low_redo"][0]) == 2
    assert np.array_equal(r0["low_components"], r1["low_components"]) and np.array_equal(r0["low_singular"], r1["low_singular"])
    sv = low.singular_values()
    assert np.allclose(r0["low_singular"][:3], sv[:3], rtol=2e-6) and np.abs(r0["low_singular"][3:]).max() <= 1e-5 * sv[0]
    assert pc.rowwise_rel(r0["low_components"][:3], low.components()[:3]).max() < 2e-5

    own = petal.RandomizedPca(k, ctx=ctx, n_iter=4, rng=np.random.default_rng(100)).fit(x)   # rank 0's generator
    assert pc.rowwise_rel(r0["own_rpca_components"], own.components()).max() < 2e-6
    assert np.allclose(r0["own_rpca_singular"], own.singular_values(), rtol=2e-6)

    x64 = synth_pca(900, 12, 3, seed=78, dtype=np.float64)
    o = po.PcaOracle(3)
    yo = o.fit_transform(x64)
    assert pc.rowwise_rel(r0["pca_components"], o.components).max() < 1e-9
    y = np.concatenate([r0["pca_y"], r1["pca_y"]])
    assert np.abs(y - yo).max() <= 1e-8 * np.abs(yo).max()      # signs included: same svd_flip as the oracle

    xi = synth_ica(4000, 5, 5, seed=79, dtype=np.float64)
    w0 = np.random.default_rng(9).standard_normal((5, 5))
    si = petal.FastIca(ctx=ctx)
    ysi = si.fit_transform(xi, w_init=w0)
    assert int(r0["ica_n_iter"][0]) == si.n_iter
    assert np.allclose(r0["ica_components"], si.components, atol=1e-9)
    assert np.allclose(np.concatenate([r0["ica_y"], r1["ica_y"]]), ysi, atol=1e-9)
    own = petal.FastIca(np.random.default_rng(200), ctx).fit(xi)
    assert int(r0["own_ica_n_iter"][0]) == own.n_iter
    assert np.allclose(r0["own_ica_components"], own.components, atol=1e-9)

"""BASELINE configs[3] and configs[4] AT THEIR STATED SIZE, sharded 8 ways, on the one MI355X of the test box.

  configs[3]  RandomizedPca k = 128 on 2 000 000 x 1024 fp32, sample-sharded across 8 ranks (n_iter = 7, src/pca.rs:680)
  configs[4]  FastIca n_components = 64 on 4 000 000 x 512 fp32, whitened-GEMM sharded 8 x, convergence tol 1e-4

Eight rank processes share GPU 0 (tests/fullsize_worker.py): each has its own ctx on libpetal_hip.so and ITS 250 000 / 500 000
rows of the one planted matrix in device memory, collective hook on a gloo group (host-staged: RCCL refuses several ranks on
one device).  The reference has no counterpart (src/pca.rs:509-550 and src/ica.rs:167-221 are single-process), so the checker
for the full 8.2 GB problems is (i) ONE single-process HIP fit of the whole matrix (the oracle sees this shard shape in
tests/test_gpu_parity.py::test_cfg4_share_against_the_oracle), (ii) the domain's size-independent properties, and (iii) the
sharding contract itself: replicated outputs bit-identical on all ranks, svd_flip (src/pca.rs:826-839) decided by elements that
live on different ranks, at most n_iter + 3 all-reduces per RandomizedPca fit."""
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

import fullsize_cases as fc
import parity_cases as pc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def ranks(tmp_path_factory):
    out = tmp_path_factory.mktemp("fullsize")
    port = _free_port()
    procs, logf = [], []
    for r in range(fc.WORLD):   # fresh children, started before this process hands them anything GPU-related
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(fc.WORLD), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        # every worker writes to its OWN file: with pipes drained one after another a later rank that fills its 64 KB pipe
        # (RCCL / HIP warnings, a traceback) blocks mid-collective while the harness waits on rank 0
        logf.append(open(os.path.join(out, f"rank{r}.log"), "w"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "fullsize_worker.py"), str(out)],
                                      env=env, stdout=logf[-1], stderr=subprocess.STDOUT))
    deadline = time.monotonic() + 1500
    try:
        for p in procs:
            p.wait(timeout=max(1.0, deadline - time.monotonic()))
    except subprocess.TimeoutExpired:
        for q in procs:
            q.kill()
        raise
    finally:
        for fh in logf:
            fh.close()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n" + open(os.path.join(out, f"rank{r}.log")).read()[-3000:]
    return [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(fc.WORLD)]


def test_replicated_outputs_are_bit_identical_on_all_eight_ranks(ranks):
    keys = [k for k in ranks[0].files if k.split(".", 1)[1] in ("components", "singular", "evr", "mean", "n_iter", "allreduce")]
    assert len(keys) == 9, keys
    for k in keys:
        assert np.all(np.isfinite(ranks[0][k])), k
        for r in ranks[1:]:
            assert np.array_equal(ranks[0][k], r[k]), k


def test_cfg4_full_size_8_shards_against_the_single_process_fit(ranks):
    import torch
    import petal_decomposition_amd as petal
    c4 = fc.CFG4
    n, d, k, rows = c4["n"], c4["d"], c4["k"], c4["n"] // fc.WORLD
    x = torch.empty((n, d), dtype=torch.float32, device="cuda")               # the whole 8.2 GB matrix, one process
    for b in range(fc.WORLD):
        fc.cfg4_block(b, out=x[b * rows:(b + 1) * rows])
        # same rows as rank b generated for itself (a checksum: the two processes ran the same device generator)
        assert abs(float(x[b * rows:(b + 1) * rows].double().sum()) - float(ranks[b]["cfg4.xsum"][0])) <= 1e-9 * n * d
    ctx = petal.Context(0)
    single = petal.RandomizedPca(k, ctx=ctx, n_iter=c4["n_iter"])
    ys = single.fit_transform(x, omega=fc.cfg4_omega())
    r0 = ranks[0]
    cm, cs = r0["cfg4.components"].astype(np.float64), single.components().astype(np.float64)
    sig, sigs = r0["cfg4.singular"].astype(np.float64), single.singular_values().astype(np.float64)
    # (1) the sharded fit IS the single-process fit: singular values to 5e-6, the leading half of the components to 2e-5 WITH
    # their signs (no sign alignment: the cross-rank svd_flip must elect the same element); the trailing half sits in 2.7 %
    # spectral gaps next to the noise floor, where fp32 determines the vectors to ~1e-7 sigma_1 / gap only
    assert np.allclose(sig, sigs, rtol=5e-6), np.abs(sig / sigs - 1).max()
    err = np.linalg.norm(cm - cs, axis=1) / np.linalg.norm(cs, axis=1)
    assert err[: k // 2].max() < 2e-5, err[: k // 2].max()
    assert pc.rowwise_rel(cm, cs).max() < 5e-3
    assert np.all(np.sign(np.sum(cm * cs, axis=1)) == 1)                        # every one of the 128 signs
    assert np.allclose(r0["cfg4.evr"], single.explained_variance_ratio(), rtol=2e-5)
    assert np.abs(r0["cfg4.mean"] - single.mean()).max() <= 1e-6 * np.abs(single.mean()).max()
    # (2) svd_flip across ranks: the first max-|.| element of each U column, and which rank's rows hold it
    colmax = np.stack([r["cfg4.colmax"] for r in ranks])                      # 8 x k
    owner = colmax.argmax(axis=0)
    assert len(set(owner.tolist())) >= 3, owner
    arg_single = ys.abs().argmax(dim=0).cpu().numpy()
    arg_sharded = np.array([ranks[owner[j]]["cfg4.colarg"][j] for j in range(k)])
    assert np.array_equal(arg_single[: k // 2], arg_sharded[: k // 2])         # the same deciding rows as the single-process fit
    # (3) size-independent properties at the full size
    assert np.abs(cm @ cm.T - np.eye(k)).max() < 2e-5                           # orthonormal rows
    assert np.all(np.diff(sig) <= 1e-6 * sig[0])                                # descending
    v, s, _ = fc.cfg4_factors()
    assert np.allclose(sig[: k // 2], s[: k // 2], rtol=0.02)                   # the planted spectrum (G is only ~orthonormal)
    cosines = np.linalg.svd(v[:, : k // 2].T @ cm.T, compute_uv=False)          # planted leading subspace inside the fitted one
    assert cosines.min() > 1 - 1e-4
    assert 0.99 < r0["cfg4.evr"].sum() <= 1.0 + 1e-5
    for r in ranks:                                                             # on every rank's own rows
        assert r["cfg4.idem"][0] < 1e-3 * r["cfg4.idem"][1]                     # transform . inverse_transform is a projection
        assert r["cfg4.y_vs_transform"][0] < 2e-3 * r["cfg4.y_vs_transform"][1]  # fit_transform == transform (pca.rs:474-480)
    # (4) the collective budget: prologue + (n_iter + 1) products + svd_flip key
    assert r0["cfg4.allreduce"][0] <= c4["n_iter"] + 3, r0["cfg4.allreduce"]
    ctx.close()


def test_cfg5_full_size_8_shards_against_the_single_process_fit(ranks):
    import torch
    import petal_decomposition_amd as petal
    c5 = fc.CFG5
    n, d, nc, rows = c5["n"], c5["d"], c5["nc"], c5["n"] // fc.WORLD
    x = torch.empty((n, d), dtype=torch.float32, device="cuda")               # 8.2 GB
    src = torch.empty((n, nc), dtype=torch.float32, device="cuda")
    for b in range(fc.WORLD):
        _, s = fc.cfg5_block(b, out=x[b * rows:(b + 1) * rows], want_sources=True)
        src[b * rows:(b + 1) * rows] = s
        assert abs(float(x[b * rows:(b + 1) * rows].double().sum()) - float(ranks[b]["cfg5.xsum"][0])) <= 1e-9 * n * d
    ctx = petal.Context(0)
    single = petal.FastIca(ctx=ctx, n_components=nc, tol=c5["tol"])
    ys = single.fit_transform(x, w_init=fc.cfg5_w0())
    r0 = ranks[0]
    it = int(r0["cfg5.n_iter"][0])
    assert 1 <= single.n_iter < 200 and abs(it - single.n_iter) <= 1, (it, single.n_iter)
    # the unmixing rows agree up to the 1e-4 stopping criterion (the sharded sums associate differently)
    w, ws = r0["cfg5.components"].astype(np.float64), single.components.astype(np.float64)
    c = w @ np.linalg.pinv(ws)
    assert np.abs(c - np.eye(nc)).max() < 2e-3, np.abs(c - np.eye(nc)).max()
    assert np.abs(r0["cfg5.mean"] - single.means).max() <= 1e-6 * max(1.0, np.abs(single.means).max())
    # the domain's property at full size: every planted Laplace source is recovered by exactly one component -- by the
    # single-process fit over all 4 000 000 samples, and by the sharded fit on every rank's own 500 000
    for corr in [fc.source_match(ys, src)] + [r["cfg5.corr"] for r in ranks]:
        assert corr.max(axis=1).min() > 0.99 and corr.max(axis=0).min() > 0.99
        assert len(set(corr.argmax(axis=1))) == nc
    # collective budget: prologue + covariance + one (nc^2 + nc) all-reduce per iteration (the flag is read every 4th) + the
    # 8-byte agreement on the optimistic whitening's verdict (every rank redoes the fit, or none: round 4)
    assert r0["cfg5.allreduce"][0] <= 3 + 4 * ((it + 3) // 4), r0["cfg5.allreduce"]
    ctx.close()

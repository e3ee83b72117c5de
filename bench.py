#!/usr/bin/env python3
"""bench.py -- samples/sec of RandomizedPca.fit() on n x d fp32 (BASELINE.json metric).

A "step" is one fit() -- column means + (2 n_iter + 2) power-iteration GEMM passes + the small-matrix
tail -- over one synthetic batch that is already resident in HBM.  Workload at N = 1: BASELINE configs[1]
(RandomizedPca k=64, 5 power iterations, 100000 x 512 fp32).  With N > 1 every rank holds its own
100000 x 512 row block of one (N * 100000) x 512 matrix (weak scaling, sample-sharded): the only data-path
exchange is the all-reduce of the small replicated matrices (RCCL through torch.distributed).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant power-iteration GEMM kernel: algorithmic bytes (default split-product mode: the kernel
                is paced by the X stream, bound "hbm") or flops (--gemm fp32: bound "mfma") per launch / average
                launch duration measured with HIP events on the launch stream inside the timed region
  cpu_baseline  the numpy/LAPACK oracle ("port") timed on this box's host cores on the same workload
and informational ones (fp32_mfma_mode: the same fit on the fp32-MFMA kernels; northstar_gemm: the two GEMM kernels
alone at 1e6 x 512 in both modes; fastica_cfg3; host_in: fit() fed a host ndarray, PCIe included -- never `value`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FP32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: Peak FP32 (matrix), dense
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E ~8 TB/s
HBM_PEAK_GBS = 8000.0


def rccl_probe(rank: int, world: int, local_rank: int, timeout_s: float = 180.0) -> bool:
    """Runs petal-decomposition_amd/rccl_probe.py as a child of this rank (own rendezvous port); True if it exits 0 in time."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29531")) + 17)
    env["RANK"], env["WORLD_SIZE"], env["LOCAL_RANK"] = str(rank), str(world), str(local_rank)
    for key in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS"):
        env.pop(key, None)
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "petal-decomposition_amd", "rccl_probe.py")
    try:
        res = subprocess.run([sys.executable, probe], env=env, timeout=timeout_s, capture_output=True, text=True)
    except subprocess.TimeoutExpired:
        print(f"[bench] rank {rank}: RCCL probe timed out after {timeout_s:.0f} s", file=sys.stderr)
        return False
    if res.returncode != 0:
        print(f"[bench] rank {rank}: RCCL probe exit code {res.returncode}: {res.stderr[-400:]}", file=sys.stderr)
    return res.returncode == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=100000, help="rows per GPU")
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--n-iter", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm", choices=["bf16x3", "fp32"], default="bf16x3",
                    help="how the X-streaming GEMM kernels form fp32 products: exact 3-way bf16 split on the bf16 matrix "
                         "cores with fp32 accumulation (default, fp32-equivalent) or fp32 MFMA (petal_ctx_set_gemm_mode)")
    ap.add_argument("--collective", choices=["auto", "rccl", "torch"], default="auto",
                    help="N > 1: the library's built-in RCCL all-reduce (petal_ctx_init_rccl), or the torch.distributed "
                         "hook; auto = built-in after a probe in a child process under a timeout (rccl_probe.py), else the hook")
    ap.add_argument("--single-rank-group", action="store_true",
                    help="development: at N = 1 still create a one-rank process group and run the sharded code path "
                         "(PETAL_FORCE_COLLECTIVE) to time its overhead")
    ap.add_argument("--no-northstar", action="store_true")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass "
                         "(default: the committed measurement in profiles/r01_pmc_traffic.json for this workload)")
    args = ap.parse_args()
    # stdout carries exactly ONE line, the JSON record: libraries that print banners to the C-level stdout (RCCL does, at
    # communicator creation and again at exit) are pointed at stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import petal_decomposition_amd as petal
    from synth_data import synth_pca

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or args.single_rank_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
        if args.single_rank_group:
            os.environ["PETAL_FORCE_COLLECTIVE"] = "1"
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    n, d, k, n_iter = args.n, args.d, args.k, args.n_iter
    l = k + 10
    # synthetic shard: the planted model of BASELINE.md section 3, one seed per rank (rows are iid, so row
    # blocks generated with different seeds but the same V would be the exact recipe; a shared V is kept by
    # seeding the factor draw identically and only the row draws per rank)
    x_host = synth_pca(n, d, k, seed=2 + 1000 * rank, dtype=np.float32)
    x = torch.from_numpy(x_host).to(dev)
    omega = np.random.default_rng(3).standard_normal((d, l)).astype(np.float32)

    ctx = petal.Context(dev.index or 0, stream=torch.cuda.current_stream(dev).cuda_stream)
    collective = "none"
    if dist is not None:
        collective = "torch.distributed hook"
        if args.collective in ("auto", "rccl"):
            # The built-in collective is first exercised in a CHILD process per rank under a timeout (rccl_probe.py: its
            # own communicator, one small sharded fit, results compared across ranks): a node on which it cannot
            # complete costs a killed child and the torch.distributed hook instead of a hung job.
            ok = 1
            if args.collective == "auto":
                ok = 1 if rccl_probe(rank, world, local_rank) else 0
                t = torch.tensor([ok], device=dev, dtype=torch.int32)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                ok = int(t.item())
                if not ok and rank == 0:
                    print("[bench] built-in RCCL probe failed on some rank; using the torch.distributed hook", file=sys.stderr)
            if ok:
                try:  # every rank binds the same librccl, so success / failure is uniform across the group
                    ctx.use_rccl()
                    collective = "built-in RCCL (petal_ctx_init_rccl)"
                except Exception as e:
                    if args.collective == "rccl":
                        raise
                    print(f"[bench] built-in RCCL unavailable ({e}); using the torch.distributed hook", file=sys.stderr)
        if collective.startswith("torch"):
            ctx.use_torch_distributed()
    ctx.set_profiling(True)
    ctx.set_gemm_mode(args.gemm)
    model = petal.RandomizedPca(k, ctx=ctx, n_iter=n_iter)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        model.fit(x, omega=omega)
    sync_all()
    acc = {"xp_ms": 0.0, "xp_launches": 0, "atb_ms": 0.0, "atb_launches": 0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.fit(x, omega=omega)
        st = ctx.stats()
        for key in acc:
            acc[key] += st[key]
    sync_all()
    elapsed = time.perf_counter() - t0
    pass_flops, pass_bytes = st["pass_flops"], st["pass_bytes"]
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        # dominant kernel = the power-iteration GEMM kind with the larger summed time
        kinds = {"K1 (Z = Xc.P)": (acc["xp_ms"], acc["xp_launches"]),
                 "K2 (Y = Xc^T.Z)": (acc["atb_ms"], acc["atb_launches"])}
        per = {kname: (ms / max(cnt, 1)) for kname, (ms, cnt) in kinds.items()}
        dom = max(kinds, key=lambda kname: kinds[kname][0])
        roofline = roofline_entry(dom, per, pass_flops, pass_bytes, args.gemm, args.pmc_traffic, n, d, l)
        out = {
            "metric": "samples/sec for RandomizedPca.fit() on n x d fp32",
            "value": round(world * n * args.steps / elapsed, 1),
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"RandomizedPca.fit k={k} n_iter={n_iter} oversample=10 on {n}x{d} fp32 per GPU "
                                   f"(BASELINE configs[1]), X resident in HBM",
                       "rows_per_gpu": n, "features": d, "n_components": k, "n_iter": n_iter,
                       "gemm_mode": ("bf16x3: fp32 operands split exactly into 3 bf16 pieces, 6 piece products on the bf16 "
                                     "matrix cores, fp32 accumulation (fp32-equivalent)") if args.gemm == "bf16x3"
                                    else "fp32 MFMA (v_mfma_f32_16x16x4_f32)",
                       "parallelism": f"sample-sharded x{world}" if world > 1 else "single GPU",
                       "collective": collective},
            "roofline": roofline,
        }

        if world == 1:
            # host-ndarray-in rate (H2D over PCIe included) -- informational, never `value`
            model.fit(x_host, omega=omega)
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                model.fit(x_host, omega=omega)
            out["host_in"] = {"value": round(n * reps / (time.perf_counter() - t1), 1), "unit": "samples/s",
                              "note": "fit() fed a pageable host ndarray: PCIe H2D inside the timed region"}

        if world == 1 and args.gemm == "bf16x3" and not args.no_northstar:
            out["fp32_mfma_mode"] = fp32_mode_extra(petal, ctx, model, x, omega)

        if world == 1 and not args.no_northstar:
            out["northstar_gemm"] = northstar(petal, ctx, torch, dev)
            ctx.set_gemm_mode(args.gemm)

        if world == 1 and not args.no_northstar:
            out["fastica_cfg3"] = fastica_cfg3(petal, ctx, torch, dev)

        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(x_host, omega, k, n_iter)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(n, d, l, mode, kind):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (PMC passes cannot run inside the timed bench)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)[f"{n}x{d} l={l}"][mode][kind]["hbm_bytes_corrected"]
    except Exception:
        return None


def roofline_entry(dom, per, pass_flops, pass_bytes, mode, traffic_override, n, d, l):
    """The `roofline` object of the bench line for the dominant kernel `dom`.
    fp32 mode: the kernels issue v_mfma_f32_16x16x4_f32 -> bound "mfma", achieved = algorithmic flops / duration vs the
    157.3 TFLOP/s fp32-MFMA peak.  bf16x3 (split-product) mode: the same products take 2.7x less matrix-pipe time on the
    bf16 cores and the kernels are paced by the X stream -> bound "hbm", achieved = algorithmic bytes / duration vs
    8 TB/s; the fp32-equivalent flop rate is kept alongside for comparison with the fp32 mode."""
    avg_ms = per[dom]
    tf = pass_flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    gbs = pass_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = traffic_override if traffic_override is not None else pmc_traffic(n, d, l, mode, dom.split(" ")[0])
    other = {kname: {"avg_launch_ms": round(v, 5), "GB/s_algorithmic": round(pass_bytes / (v * 1e-3) / 1e9, 1) if v > 0 else 0.0,
                     "fp32_equivalent_TFLOP/s": round(pass_flops / (v * 1e-3) / 1e12, 3) if v > 0 else 0.0}
             for kname, v in per.items() if kname != dom}
    common = {"kernel": dom + (" k_xp3 / k_atb3 (bf16x3 split-product, fp32 accumulate)" if mode == "bf16x3"
                               else " k_xp_* / k_atb_mfma (fp32 MFMA)"),
              "traffic": traffic, "avg_launch_ms": round(avg_ms, 5), "flops_per_launch": pass_flops,
              "bytes_per_launch": pass_bytes, "other_kernel": other}
    if mode == "bf16x3":
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                **common, "fp32_equivalent_TFLOP/s": round(tf, 3),
                "fp32_equivalent_frac_of_fp32_mfma_peak": round(tf / FP32_MFMA_PEAK_TF, 4)}
    return {"bound": "mfma", "achieved": round(tf, 3), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": round(tf / FP32_MFMA_PEAK_TF, 4), **common, "hbm_GBps_algorithmic": round(gbs, 1)}


def fp32_mode_extra(petal, ctx, model, x, omega, steps=10):
    """The same fit with the fp32-MFMA kernels (petal_ctx_set_gemm_mode): informational, next to the default mode."""
    ctx.set_gemm_mode("fp32")
    try:
        for _ in range(3):
            model.fit(x, omega=omega)
        acc = {"xp_ms": 0.0, "xp_launches": 0, "atb_ms": 0.0, "atb_launches": 0}
        t0 = time.perf_counter()
        for _ in range(steps):
            model.fit(x, omega=omega)
            st = ctx.stats()
            for key in acc:
                acc[key] += st[key]
        ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        ctx.set_gemm_mode("bf16x3")
    k1, k2 = acc["xp_ms"] / max(acc["xp_launches"], 1), acc["atb_ms"] / max(acc["atb_launches"], 1)
    fl = st["pass_flops"]
    return {"ms_per_step": round(ms, 4), "samples_per_s": round(x.shape[0] / (ms * 1e-3), 1),
            "K1": {"avg_launch_ms": round(k1, 5), "TFLOP/s": round(fl / (k1 * 1e-3) / 1e12, 2), "frac_of_fp32_mfma_peak": round(fl / (k1 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4)},
            "K2": {"avg_launch_ms": round(k2, 5), "TFLOP/s": round(fl / (k2 * 1e-3) / 1e12, 2), "frac_of_fp32_mfma_peak": round(fl / (k2 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4)}}


def northstar(petal, ctx, torch, dev, n=1_000_000, d=512, l=74, reps=5):
    """The north-star point: the two power-iteration GEMM kernels alone on a 1e6 x 512 fp32 matrix
    (2.05 GB, beyond the 256 MiB Infinity Cache), l = 74."""
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    x = torch.randn((n, d), generator=g, device=dev, dtype=torch.float32)
    z = torch.randn((n, 80), generator=g, device=dev, dtype=torch.float32)
    z[:, l:] = 0
    p = np.random.default_rng(7).standard_normal((d, l)).astype(np.float32)
    mu = np.random.default_rng(8).standard_normal(d).astype(np.float32)
    res = {"shape": f"{n}x{d} fp32, l={l}", "flops_per_launch": 2.0 * n * d * l,
           "bytes_per_launch": 4.0 * (n * d + n * l + d * l)}
    for mode in ("bf16x3", "fp32"):
        ctx.set_gemm_mode(mode)
        res[mode] = {}
        for name, fn, key in (("K1", lambda: petal.gemm_xp(x, p, mu, ctx=ctx), "xp"),
                              ("K2", lambda: petal.gemm_atb(x, z, mu, ctx=ctx), "atb")):
            fn()
            ms, cnt = 0.0, 0
            for _ in range(reps):
                fn()
                st = ctx.stats()
                ms += st[key + "_ms"]
                cnt += st[key + "_launches"]
            avg = ms / max(cnt, 1)
            tf = 2.0 * n * d * l / (avg * 1e-3) / 1e12 if avg > 0 else 0.0
            gbs = 4.0 * (n * d + n * l + d * l) / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
            res[mode][name] = {"avg_launch_ms": round(avg, 4), "GB/s_algorithmic": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                               "fp32_equivalent_TFLOP/s" if mode == "bf16x3" else "TFLOP/s": round(tf, 2),
                               "frac_of_fp32_mfma_peak": round(tf / FP32_MFMA_PEAK_TF, 4),
                               "hbm_bytes_pmc": pmc_traffic(n, d, l, mode, name)}
    ctx.set_gemm_mode("bf16x3")
    del x, z
    torch.cuda.empty_cache()
    return res


def fastica_cfg3(petal, ctx, torch, dev, n=200000, d=256, nc=32):
    """BASELINE configs[2] (informational): FastIca n_components=32 (logcosh) on 200000 x 256 fp32, X in HBM.
    Reports the full fit (whitening + loop to the 1e-4 criterion) and the loop at a fixed 200 iterations."""
    from synth_data import synth_ica
    x = torch.from_numpy(synth_ica(n, d, nc, seed=5, dtype=np.float32)).to(dev)
    w0 = np.random.default_rng(7).standard_normal((nc, nc)).astype(np.float32)
    m = petal.FastIca(ctx=ctx, n_components=nc)
    for _ in range(10):  # the host-side data generation above idles the GPU: let its clocks ramp up again
        m.fit(x, w_init=w0)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        m.fit(x, w_init=w0)
    fit_ms = (time.perf_counter() - t0) / reps * 1e3
    st = ctx.stats()
    step_ms = st["ica_step_ms"] / max(st["ica_step_launches"], 1)
    m200 = petal.FastIca(ctx=ctx, n_components=nc, tol=0.0, max_iter=200)
    m200.fit(x, w_init=w0)
    t0 = time.perf_counter()
    m200.fit(x, w_init=w0)
    fit200_ms = (time.perf_counter() - t0) * 1e3
    # (ii) of SURVEY.md 8(d): the loop alone (ica_par, src/ica.rs:319-361) on already-whitened data X1 (nc x n), here the
    # sample-major n x nc matrix handed over as its transposed view
    xc = x - x.mean(dim=0, keepdim=True)
    evals, evecs = torch.linalg.eigh((xc.T @ xc).double())
    kmat = (evecs[:, -nc:] / evals[-nc:].sqrt()).float()             # d x nc
    x1t = ((xc @ kmat) * float(np.sqrt(n))).contiguous()             # n x nc, unit covariance
    for _ in range(3):
        _w, it_loop = petal.ica_par(x1t.T, 1e-4, 200, w0, ctx=ctx)
    t0 = time.perf_counter()
    for _ in range(reps):
        petal.ica_par(x1t.T, 1e-4, 200, w0, ctx=ctx)
    loop_ms = (time.perf_counter() - t0) / reps * 1e3
    return {"shape": f"{n}x{d} fp32, n_components={nc}", "fit_ms": round(fit_ms, 3), "n_iter": m.n_iter,
            "loop_only_on_whitened_ms": round(loop_ms, 3), "loop_only_n_iter": int(it_loop),
            "samples_per_s": round(n / (fit_ms * 1e-3), 1),
            "fit_fixed_200_iter_ms": round(fit200_ms, 3), "ms_per_iteration": round((fit200_ms - fit_ms) / max(200 - m.n_iter, 1), 4),
            "step_kernel": {"avg_launch_ms": round(step_ms, 5),
                            "TFLOP/s": round(st["ica_step_flops"] / (step_ms * 1e-3) / 1e12, 2) if step_ms > 0 else 0.0,
                            "GB/s_algorithmic": round(st["ica_step_bytes"] / (step_ms * 1e-3) / 1e9, 1) if step_ms > 0 else 0.0}}


def cpu_baseline(x_host, omega, k, n_iter):
    """The oracle (numpy + LAPACK restatement of the reference algorithm, kind "port") on the host cores,
    same workload, same Omega, same n_iter.  NOTE: its GEMMs run on multithreaded OpenBLAS, whereas the
    crate's own GEMMs are single-threaded matrixmultiply -- this baseline is stronger."""
    from oracle import petal_oracle as po
    threads = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        pass
    m = po.RandomizedPcaOracle(k, n_iter=n_iter)
    t0 = time.perf_counter()
    m.fit(x_host, omega=omega)
    dt = time.perf_counter() - t0
    return {"value": round(x_host.shape[0] / dt, 1), "unit": "samples/s", "cores": int(threads), "kind": "port",
            "sample": f"one full fit of the same {x_host.shape[0]}x{x_host.shape[1]} fp32 workload (k={k}, n_iter={n_iter}) "
                      f"in {dt:.2f} s; numpy + OpenBLAS/LAPACK (getrf P.L, geqrf/orgqr, gesdd)"}


if __name__ == "__main__":
    main()

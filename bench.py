#!/usr/bin/env python3
"""bench.py -- samples/sec of RandomizedPca.fit() on n x d fp32 (BASELINE.json metric).

A "step" is one fit() -- column means + (2 n_iter + 2) power-iteration GEMM passes + the small-matrix
tail -- over one synthetic batch that is already resident in HBM.  Workload at N = 1: BASELINE configs[1]
(RandomizedPca k=64, 5 power iterations, 100000 x 512 fp32), the configuration the metric is quoted on.
With N > 1 (and no --config) the workload is the one north_star scales: BASELINE configs[3], ONE 2000000 x 1024 matrix,
k = 128, 7 power iterations (the crate's default, src/pca.rs:679-680), its rows split N ways -- STRONG scaling,
`value` = 2000000 samples / time per fit; `--gpus 1 --config cfg4s` is the same matrix on one GPU (8.2 GB: it fits).
`--config cfg2 | cfg4 | cfg5 --gpus N` keep the weak-scaling lines (every rank its own block of the per-GPU size).
The only data-path exchange is the all-reduce of the small replicated matrices (RCCL); the record carries
`allreduce_calls`, `allreduce_bytes` and the stream time spent in them per fit.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant power-iteration GEMM kernel: algorithmic bytes (default split-product mode: the kernel
                is paced by the X stream, bound "hbm") or flops (--gemm fp32: bound "mfma") per launch / average
                launch duration measured with HIP events on the launch stream inside the timed region
  cpu_baseline  the numpy/LAPACK oracle ("port") timed on this box's host cores on the same workload
and informational ones (fp32_mfma_mode: the same fit on the fp32-MFMA kernels; northstar_gemm: the two GEMM kernels
alone at 1e6 x 512 in both modes; fastica_cfg3; pca_cfg1: the exact Pca on configs[0]; host_in: fit() fed a host ndarray, PCIe
included -- never `value`).
"""
import argparse
import datetime
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


CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on (one GPU's share under weak scaling)
    "cfg2": dict(model="rpca", n=100000, d=512, k=64, n_iter=5, label="BASELINE configs[1]"),
    # BASELINE.json configs[3]: 2000000 x 1024, k = 128, sample-sharded over 8 GPUs -> 250000 rows per rank, crate default n_iter
    "cfg4": dict(model="rpca", n=250000, d=1024, k=128, n_iter=7, label="BASELINE configs[3], one rank's 1/8 share per GPU"),
    # BASELINE.json configs[4]: FastIca n_components = 64 on 4000000 x 512 over 8 GPUs -> 500000 rows per rank, tol 1e-4
    "cfg5": dict(model="ica", n=500000, d=512, k=64, n_iter=0, label="BASELINE configs[4], one rank's 1/8 share per GPU"),
    # BASELINE.json configs[3] as north_star scales it: ONE 2000000 x 1024 matrix split over the N ranks (strong scaling); `n` is
    # the TOTAL row count here (rows per GPU = n / N), the data is generated on the device (synth_data.synth_pca_device)
    "cfg4s": dict(model="rpca", n=2000000, d=1024, k=128, n_iter=7, strong=True, seed=4,
                  label="BASELINE configs[3]: one 2000000x1024 matrix sample-sharded over the GPUs, strong scaling"),
}


def default_config(gpus: int) -> str:
    """N = 1: the configuration the metric is quoted on (configs[1]); N > 1: the configuration north_star scales (configs[3], strong)"""
    return "cfg2" if gpus <= 1 else "cfg4s"


def shard_rows(n_total: int, world: int, rank: int):
    """[begin, end) of rank's row block when n_total rows are split over world ranks (the first n_total % world ranks get one more)"""
    base, extra = divmod(n_total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def launch_ranks(n_ranks: int, share_gpu: bool = False) -> int:
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU) BEFORE this process touches a
    GPU -- fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, never a re-exec of a process that has
    initialised HIP -- wait for them and return the worst exit code.  Rank 0 inherits stdout (the one JSON line)."""
    import socket
    import subprocess
    import torch  # torch.cuda.device_count() does not initialise the GPU on this image
    visible = torch.cuda.device_count()
    if visible < (1 if share_gpu else n_ranks):
        print(f"[bench] --gpus {n_ranks} needs {n_ranks} visible GPUs, this node shows {visible}: refusing to run "
              f"(a {n_ranks}-GPU line measured on fewer GPUs would be wrong)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if share_gpu else str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    # (ALL ranks are watched, not rank 0 first: a rank that dies leaves the others waiting in a collective for ever, and a parent
    # blocked in rank 0's wait() would never see it)
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            if p.poll() is None:
                continue
            live.remove(p)
            rc = max(rc, abs(p.returncode))
            if p.returncode != 0:  # end exactly the ones we started
                print(f"[bench] a rank process exited with {p.returncode}: ending the others", file=sys.stderr)
                for q in live:
                    q.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="default: cfg2 = BASELINE configs[1] at --gpus 1 (the metric's configuration); cfg4s = BASELINE configs[3], "
                         "one 2000000 x 1024 matrix split over the GPUs (strong scaling), at --gpus N > 1.  cfg2 / cfg4 / cfg5 with "
                         "--gpus N: per-GPU workloads (weak scaling); cfg4 / cfg5 = one rank's share of the 8-GPU configs[3] / configs[4]")
    ap.add_argument("--n", type=int, default=None, help="rows per GPU (cfg4s: rows of the whole matrix); default: the config's")
    ap.add_argument("--d", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--n-iter", type=int, default=None)
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
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing only: the N ranks all use GPU 0 and all-reduce through a gloo group (the hook stages the small "
                         "buffers through the host; RCCL refuses two ranks on one device).  Exercises the launcher and the "
                         "multi-rank HIP path on a one-GPU box; the record is labelled and is not a scaling measurement")
    ap.add_argument("--no-northstar", action="store_true")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass "
                         "(default: the committed measurement in profiles/r01_pmc_traffic.json for this workload)")
    ap.add_argument("--clock-ramp-s", type=float, default=0.25,
                    help="seconds of untimed fits in front of the warm-up steps (brings the GPU clocks up after the host-side data generation)")
    ap.add_argument("--no-strong-baseline", action="store_true",
                    help="cfg4s at N > 1: skip the single-GPU fit of the whole matrix that rank 0 times after the sharded run")
    args = ap.parse_args()
    if args.config is None:
        args.config = default_config(args.gpus)
    cfg = CONFIGS[args.config]
    for key in ("n", "d", "k", "n_iter"):
        if getattr(args, key) is None:
            setattr(args, key, cfg[key])
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, args.share_gpu))   # this process never touches a GPU: its children are the ranks
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: launch one rank per GPU "
              f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or drop "
              f"WORLD_SIZE and let bench.py start the ranks itself", file=sys.stderr)
        sys.exit(2)
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
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
            args.collective = "torch"
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=600))   # (a stuck collective is an error after ten minutes, not a hung job)
        if args.single_rank_group:
            os.environ["PETAL_FORCE_COLLECTIVE"] = "1"
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    n, d, k, n_iter = args.n, args.d, args.k, args.n_iter
    l = k + 10
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
            if args.share_gpu:
                collective = "torch.distributed hook on gloo, host-staged (ranks SHARE GPU 0: launcher / path test, not a scaling point)"
    ctx.set_profiling(True)
    ctx.set_gemm_mode(args.gemm)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if cfg["model"] == "ica":
        out = bench_fastica(args, cfg, petal, ctx, torch, dist, dev, rank, world, collective, sync_all)
        if world > 1:
            out["speedup_vs_one_gpu_same_matrix"] = None   # (a weak-scaling line: no same-matrix one-GPU fit in this job; the key is on every N > 1 line)
        if rank == 0:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    strong = bool(cfg.get("strong"))
    n_total = n if strong else world * n
    if strong:
        # ONE planted matrix, defined block by block on the device: rank r holds rows [begin, end) of it
        from synth_data import synth_pca_device
        row_begin, row_end = shard_rows(n_total, world, rank)
        x_host = None
        x = synth_pca_device(n_total, d, k, cfg["seed"], row_begin, row_end, dev)
        n = row_end - row_begin
    else:
        # synthetic shard: the planted model of BASELINE.md section 3; V and mu are shared by all ranks (seed 2), the iid rows of
        # each rank's block come from its own seed
        x_host = synth_pca(n, d, k, seed=2, dtype=np.float32, row_seed=None if world == 1 else 2 + 1000 * rank)
        x = torch.from_numpy(x_host).to(dev)
    omega = np.random.default_rng(3).standard_normal((d, l)).astype(np.float32)
    model = petal.RandomizedPca(k, ctx=ctx, n_iter=n_iter)

    # Clock ramp (untimed, like the W warm-up steps behind it): the seconds of host-side data generation above leave the GPU in its
    # idle clock state, and W = 5 one-millisecond fits do not bring it back (measured: the first 25 ms after an idle period run
    # 5-10 % slow).  A quarter of a second of the same fits does; the timed region below is still exactly K steps.
    ramp(lambda: model.fit(x, omega=omega), args.clock_ramp_s, dist, torch, "cpu" if args.share_gpu else dev)
    for _ in range(args.warmup):
        model.fit(x, omega=omega)
    sync_all()
    acc = {"xp_ms": 0.0, "xp_launches": 0, "atb_ms": 0.0, "atb_launches": 0, "pow_ms": 0.0, "pow_launches": 0,
           "allreduce_ms": 0.0, "allreduce_timed": 0}
    redo = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.fit(x, omega=omega)
        st = ctx.stats()
        for key in acc:
            acc[key] += st[key]
        redo = max(redo, int(st["rpca_redo"]))
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if args.share_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # every rank's own view of the collective (gathered: a SCALE line can then show N distinct devices inside ONE communicator)
    me = {"rank": rank, **ctx.collective_info()}
    try:
        props = torch.cuda.get_device_properties(dev)
        me["device_uuid"] = str(getattr(props, "uuid", ""))
    except Exception:
        me["device_uuid"] = None
    ranks_info = [me]
    if dist is not None:
        try:
            ranks_info = [None] * world
            dist.all_gather_object(ranks_info, me)
        except Exception as e:  # informational
            ranks_info = [me, {"error": repr(e)}]

    # (the chain's extra fits are SHARDED fits where world > 1 -- a weak-scaling run, `--config cfg2 / cfg4 --gpus N`: every rank has to run
    # them, not rank 0 alone beside ranks already waiting at the last barrier; round 6)
    chain = serial_chain(ctx, model, x, omega, elapsed / args.steps * 1e3) if (not strong or world == 1) else None
    out = None
    if rank == 0:
        # dominant kernel = the power-iteration GEMM kind with the larger summed time
        sampled = {"K1 (Z = Xc.P)": (acc["xp_ms"], acc["xp_launches"]),
                   "K2 (Y = Xc^T.Z)": (acc["atb_ms"], acc["atb_launches"]),
                   "K3 (Y' = Xc^T.(Xc.P), fused)": (acc["pow_ms"], acc["pow_launches"])}
        per = {kname: (ms / max(cnt, 1)) for kname, (ms, cnt) in sampled.items()}
        # (level-1 profiling samples ONE launch per fit, the kind rotating: the SUMS over a few steps say which kind was sampled
        # more often, not which is slower.  The dominant kernel is the kind a fit spends most of its time in: the fused pass where
        # it runs -- n_iter + 1 launches per fit, K1 / K2 then do not run in the loop at all -- else the longer of K1 / K2)
        steer = steering_passes(n, d, l, n_iter, args.gemm, redo, world)
        kinds = kind_table(n, d, l, args.gemm, two_plane=(redo == 0 and n_iter > 0), k3_pieces_avg=k3_pieces(n_iter, steer), n_iter=n_iter,
                           steer12=steering_k12(l, n_iter, args.gemm, redo))
        dom = "K3 (Y' = Xc^T.(Xc.P), fused)" if per["K3 (Y' = Xc^T.(Xc.P), fused)"] > 0 else max(per, key=per.get)
        roofline = roofline_entry(dom, per, kinds, args.gemm, args.pmc_traffic, n, d, l)
        out = {
            "metric": "samples/sec for RandomizedPca.fit() on n x d fp32",
            "value": round(n_total * args.steps / elapsed, 1),
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"RandomizedPca.fit k={k} n_iter={n_iter} oversample=10 on ONE {n_total}x{d} fp32 matrix, rows split over "
                                    f"{world} GPU(s) ({cfg['label']}), X resident in HBM" if strong else
                                    f"RandomizedPca.fit k={k} n_iter={n_iter} oversample=10 on {n}x{d} fp32 per GPU "
                                    f"({cfg['label']}), X resident in HBM")
                                   + (" and small enough (%.0f MB) to stay in the 256 MiB Infinity Cache between passes" % (4e-6 * n * d)
                                      if 4 * n * d + 8 * n * l <= INFINITY_CACHE_BYTES else ""),
                       "x_fits_infinity_cache": bool(4 * n * d + 8 * n * l <= INFINITY_CACHE_BYTES),
                       "rows_per_gpu": n, "rows_total": n_total, "features": d, "n_components": k, "n_iter": n_iter,
                       "gemm_mode": ("bf16x3: fp32 operands split exactly into 3 bf16 pieces, 6 piece products on the bf16 "
                                     "matrix cores, fp32 accumulation (fp32-equivalent)") if args.gemm == "bf16x3"
                                    else "fp32 MFMA (v_mfma_f32_16x16x4_f32)",
                       "parallelism": f"sample-sharded x{world}" if world > 1 else "single GPU",
                       "collective": collective, "clock_ramp_s": args.clock_ramp_s},
            "roofline": roofline,
            "fit_roofline": fit_roofline(n, d, l, n_iter, 4, args.gemm, elapsed / args.steps * 1e3, steer),
            "rpca_redo": redo,   # 0: every timed fit stood on its optimistic (two-plane, fused) run
            # what went through the collective per fit (sharded runs; zeros on one GPU).  The stream time is sampled: one
            # bracketed all-reduce per fit, the index rotating from fit to fit -> average call time x calls per fit
            "collective": {**collective_entry(st, acc, args.steps, ctx, torch, dev), "ranks": ranks_info},
        }

        if per["K3 (Y' = Xc^T.(Xc.P), fused)"] > 0:
            v = per["K3 (Y' = Xc^T.(Xc.P), fused)"]
            out["fused_pass"] = {"kernel": "k_pow3", "avg_launch_ms": round(v, 5), "launches_per_fit": fused_launches(n_iter),
                                 "algorithmic_bytes": 4.0 * (n * d + 2 * d * l), "GB/s_algorithmic": round(4.0 * (n * d + 2 * d * l) / (v * 1e-3) / 1e9, 1),
                                 "note": "Y' = Xc^T (Xc P): both products of a power iteration in one pass, X read ONCE, Z neither written "
                                         "nor read (the last pass of a fit also stores Z)"}
        if chain is not None:
            out["serial_chain"] = chain
            if cfg["model"] == "rpca" and world == 1:
                out["predicted_scaling"] = predicted_scaling(out["ms_per_step"], out["serial_chain"], n_iter, d, l, strong)
        if world == 1 and x_host is not None:
            # host-ndarray-in rate (H2D over PCIe included) -- informational, never `value`
            model.fit(x_host, omega=omega)
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                model.fit(x_host, omega=omega)
            st_h = ctx.stats()
            out["host_in"] = {"value": round(n * reps / (time.perf_counter() - t1), 1), "unit": "samples/s",
                              "row_pitch_bytes": int(st_h["x_row_pitch_bytes"]), "natural_pitch_bytes": 4 * d,
                              "note": "fit() fed a pageable host ndarray: PCIe H2D inside the timed region; the copy lands with a "
                                      "padded device row pitch when the natural one is a multiple of 1 KiB (memory-channel spread)"}

        if world == 1 and not strong and not args.no_northstar:
            out["x_padded_pitch"] = padded_pitch_extra(petal, ctx, torch, model, x, omega)

        if world == 1 and args.gemm == "bf16x3" and not args.no_northstar:
            out["fp32_mfma_mode"] = fp32_mode_extra(petal, ctx, model, x, omega)
            # the two figures that price the optimism of `value`: the same fit with three-plane operands in every pass
            # (no steering passes, no verdict), and a fit whose spectrum the verdict refuses -- the caller then pays for two runs
            out["exact_mode"] = exact_mode_extra(petal, ctx, model, x, omega)
            out["redo_case"] = redo_case_extra(petal, ctx, torch, dev, n, d, k, n_iter)

        if world == 1 and not args.no_northstar and not strong:
            del x
            torch.cuda.empty_cache()
            out["northstar_gemm"] = northstar(petal, ctx, torch, dev)
            ctx.set_gemm_mode(args.gemm)
            out["northstar_fit"] = northstar_fit(petal, ctx, torch, dev, args.gemm)

        if world == 1 and not args.no_northstar and not strong:
            out["fastica_cfg3"] = fastica_cfg3(petal, ctx, torch, dev)
            out["pca_cfg1"] = pca_cfg1(petal, ctx, torch, dev)

        if world == 1 and not args.no_cpu_baseline and x_host is not None:
            out["cpu_baseline"] = cpu_baseline(x_host, omega, k, n_iter)
    if strong and world > 1 and not args.no_strong_baseline:
        # the same matrix on ONE GPU, timed by rank 0 alone after the sharded run (the other ranks wait at the barrier below):
        # the denominator of the strong-scaling ratio, measured in the same job on the same box
        del x
        torch.cuda.empty_cache()
        if rank == 0:
            try:
                out["one_gpu_same_matrix"] = strong_baseline(petal, torch, dev, cfg, n_total, d, k, n_iter, omega, args.gemm,
                                                             out["ms_per_step"])
                # top level: value(N) / value(1) across bench lines compares DIFFERENT workloads (--gpus 1 is configs[1]); this is the
                # ratio a scaling table wants -- the same matrix, the same job, one GPU against N
                out["speedup_vs_one_gpu_same_matrix"] = out["one_gpu_same_matrix"].get("speedup_of_the_sharded_run")
            except Exception as e:  # informational: never loses the line
                out["one_gpu_same_matrix"] = {"error": repr(e)}
                out["speedup_vs_one_gpu_same_matrix"] = None
    if rank == 0 and out is not None and world > 1 and "speedup_vs_one_gpu_same_matrix" not in out:
        out["speedup_vs_one_gpu_same_matrix"] = None   # (weak-scaling configs / --no-strong-baseline: no same-matrix one-GPU fit in this job)
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def ramp(fn, seconds=0.25, dist=None, torch=None, dev=None):
    """untimed repetitions of fn for `seconds`: brings the GPU clocks back up after an idle (host-side) stretch.
    With a process group (fn is then a SHARDED fit, full of all-reduces) the decision to go on is rank 0's, broadcast before every
    repetition: a loop on each rank's own clock lets two ranks disagree about the number of fits by one -- the first then sits in its
    last fit's all-reduce and the second in the barrier behind the loop, for ever (round 6: seen twice in ~20 two-rank runs)."""
    t0 = time.perf_counter()
    while True:
        go = time.perf_counter() - t0 < seconds
        if dist is not None:
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=dev)
            dist.broadcast(flag, src=0)
            go = bool(int(flag.item()))
        if not go:
            return
        fn()


def collective_entry(st, acc, steps, ctx=None, torch=None, dev=None):
    calls, timed = int(st["allreduce_calls"]), int(acc["allreduce_timed"])
    avg = acc["allreduce_ms"] / timed if timed else 0.0
    who = {}
    if ctx is not None:
        # what the communicator ITSELF reports (ncclCommCount / ncclCommCuDevice / ncclCommUserRank), not the launcher's environment:
        # a scaling line can show that RCCL spanned N ranks
        who = {"communicator": ctx.collective_info()}
        try:
            props = torch.cuda.get_device_properties(dev)
            who["device"] = {"name": props.name, "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None)}
        except Exception:
            pass
    return {**who, "allreduce_calls_per_fit": calls, "allreduce_bytes_per_fit": float(st["allreduce_bytes"]),
            "allreduce_avg_call_ms": round(avg, 5), "allreduce_ms_per_fit": round(avg * calls, 4),
            "timed_calls": timed,
            "note": "stream time between events around the all-reduce calls (includes waiting for the slowest rank)"}


def strong_baseline(petal, torch, dev, cfg, n_total, d, k, n_iter, omega, gemm, sharded_ms, reps=3):
    from synth_data import synth_pca_device
    ctx1 = petal.Context(dev.index or 0, stream=torch.cuda.current_stream(dev).cuda_stream)
    ctx1.set_gemm_mode(gemm)
    x = synth_pca_device(n_total, d, k, cfg["seed"], 0, n_total, dev)
    m = petal.RandomizedPca(k, ctx=ctx1, n_iter=n_iter)
    ramp(lambda: m.fit(x, omega=omega), 0.2)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        m.fit(x, omega=omega)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / reps * 1e3
    del x
    ctx1.close()
    return {"ms_per_step": round(ms, 4), "samples_per_s": round(n_total / (ms * 1e-3), 1),
            "speedup_of_the_sharded_run": round(ms / sharded_ms, 3) if sharded_ms > 0 else None,
            "note": f"RandomizedPca.fit of the whole {n_total}x{d} matrix on rank 0's GPU alone, {reps} fits after one warm-up"}


def pmc_traffic(n, d, l, mode, kind, with_commit=False):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (PMC passes cannot run inside the timed bench): the
    newest profiles/rNN_pmc_traffic.json that holds this shape; `with_commit` also returns the commit the kernels were at
    when it was measured ("measured_at" in the file), so a stale figure is visible as such."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
            val = doc[f"{n}x{d} l={l}"][mode][kind]["hbm_bytes_corrected"]
            return (val, doc.get("measured_at", os.path.basename(path))) if with_commit else val
        except Exception:
            continue
    return (None, None) if with_commit else None


BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: Peak BF16 MFMA, dense (never the 2:1-sparsity figure)
INFINITY_CACHE_BYTES = 256 * 1024 * 1024


def pass_bound(pass_flops, pass_bytes, mode, pieces=6.0):
    """Which roofline binds ONE power-iteration GEMM launch of this shape -- computed, not assumed: the largest of
      hbm        algorithmic bytes / 8 TB/s
      mfma-bf16  (bf16x3 mode) `pieces` bf16 piece products per fp32 product -- six, or the FIVE a launch issues when its small
                 operand is a two-plane one (K1 behind a re-basing, K1 with the sketch matrix of an optimistic fit), or 5.5 for the
                 fused pass (five in its first product, six in its second): pieces x algorithmic flops / 2.5 PFLOP/s dense
      mfma-fp32  (fp32 mode) algorithmic flops / 157.3 TFLOP/s
    Returns {"bound": "hbm" | "mfma", "pipe": ..., "floor_s": the binding time, "times_s": every candidate}."""
    times = {"hbm": pass_bytes / (HBM_PEAK_GBS * 1e9)}
    if mode == "bf16x3":
        times["mfma-bf16"] = pieces * pass_flops / (BF16_MFMA_PEAK_TF * 1e12)
    else:
        times["mfma-fp32"] = pass_flops / (FP32_MFMA_PEAK_TF * 1e12)
    pipe = max(times, key=times.get)
    return {"bound": "hbm" if pipe == "hbm" else "mfma", "pipe": pipe, "floor_s": times[pipe], "times_s": times}


def steering_passes(n, d, l, n_iter, mode, redo=0, world=1):
    """How many of a fit's n_iter + 1 fused passes are STEERING passes (k_pow3f: Xc, z and P on two bf16 planes each, FOUR piece products
    per tile in both products) -- every fused pass but the last one of the fit, which stores Z and keeps its five / six piece products."""
    if not (mode == "bf16x3" and d == 512 and l <= 80 and n_iter >= 3 and redo == 0) or os.environ.get("PETAL_NO_POW3_FAST"):
        return 0
    return fused_launches(n_iter) - 1   # (the first pass too, with or without the means gathered in it)


def fused_launches(n_iter):
    """fused passes per fit where the fused kernel runs: one per product pair, n_iter + 1 -- but n_iter at n_iter <= 4, whose first pair is
    K1, a re-basing of the sketch on the tall side, K2 (algo.cpp, `rebase_sketch`)"""
    return n_iter + 1 if n_iter >= 5 else n_iter


def k3_pieces(n_iter, steering):
    """bf16 piece products per fp32 product, averaged over the n_iter + 1 fused passes of a fit: 4 in a steering pass, 5.5 in the others"""
    return (4.0 * steering + 5.5 * (fused_launches(n_iter) - steering)) / fused_launches(n_iter)


def steering_k12(l, n_iter, mode, redo=0):
    """Do the K1 / K2 launches of a fit's loop run as STEERING products (four piece products: Xc, Z and P on two bf16 planes each)?  The
    forms for more than 80 columns do, in every iteration but the last K1 and the final K2 (whose results the fit's outputs are made of)."""
    return mode == "bf16x3" and l > 80 and n_iter >= 3 and redo == 0 and not os.environ.get("PETAL_NO_POW3_FAST")


def kind_table(n, d, l, mode, two_plane, k3_pieces_avg=5.5, n_iter=0, steer12=False):
    """algorithmic work of ONE launch of each kernel kind: (flops, bytes, bf16 piece products per fp32 product -- averaged over the
    fit's launches of the kind where steering launches issue four and the last one five or six)"""
    gemm = (2.0 * n * d * l, 4.0 * (n * d + n * l + d * l))
    k1 = 5.0 if two_plane and mode == "bf16x3" else 6.0
    k2 = 6.0
    if steer12:
        k1 = (4.0 * n_iter + k1) / (n_iter + 1)
        k2 = (4.0 * n_iter + k2) / (n_iter + 1)
    return {"K1 (Z = Xc.P)": gemm + (k1,),
            "K2 (Y = Xc^T.Z)": gemm + (k2,),
            # both products of a power iteration in ONE pass: X is read once, Z is neither written nor read
            "K3 (Y' = Xc^T.(Xc.P), fused)": (4.0 * n * d * l, 4.0 * (n * d + 2 * d * l), k3_pieces_avg)}


KERNEL_NAMES = {"K1": "k_xp3", "K2": "k_atb3", "K3": "k_pow3f / k_pow3"}


def roofline_entry(dom, per, kinds, mode, traffic_override, n, d, l):
    """The `roofline` object of the bench line for the dominant kernel `dom`, against the roofline that BINDS its shape
    (pass_bound): "hbm" -> achieved = algorithmic bytes / duration vs 8 TB/s; "mfma" on the bf16 pipe -> achieved = the bf16
    piece products' flops / duration vs 2.5 PFLOP/s dense (the fused pass at l = 74 and every kernel at l = 138: the matrix pipe
    binds, not the X stream); "mfma" on the fp32 pipe (--gemm fp32) -> algorithmic flops / duration vs 157.3 TFLOP/s.
    frac = floor time / measured duration.  `kinds`: kind_table()."""
    avg_ms = per[dom]
    flops, nbytes, pieces = kinds[dom]
    tf = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    gbs = nbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    b = pass_bound(flops, nbytes, mode, pieces)
    short = dom.split(" ")[0]
    traffic, measured_at = (traffic_override, "--pmc-traffic") if traffic_override is not None else \
        pmc_traffic(n, d, l, mode, short, with_commit=True)
    other = {}
    for kname, v in per.items():
        if kname == dom or v <= 0:
            continue
        f2, b2, p2 = kinds[kname]
        other[kname] = {"avg_launch_ms": round(v, 5), "GB/s_algorithmic": round(b2 / (v * 1e-3) / 1e9, 1),
                        "fp32_equivalent_TFLOP/s": round(f2 / (v * 1e-3) / 1e12, 3), "piece_products": p2,
                        "frac_of_binding_roofline": round(pass_bound(f2, b2, mode, p2)["floor_s"] / (v * 1e-3), 4)}
    common = {"pipe": b["pipe"],
              "candidate_floors_us": {k2: round(v * 1e6, 2) for k2, v in b["times_s"].items()},
              "kernel": dom + (f" {KERNEL_NAMES.get(short, '')} (bf16x3 split-product, fp32 accumulate)" if mode == "bf16x3"
                               else " k_xp_* / k_atb_mfma (fp32 MFMA)"),
              "piece_products": pieces if mode == "bf16x3" else None,
              "traffic": traffic, "traffic_measured_at": measured_at,
              "traffic_note": "FETCH_SIZE x 2 + WRITE_SIZE (the gfx950 guide's correction); the x 2 is exact for 128-B requests but "
                              "over-counts reads issued as 64-B segments such as K2's Z stage (factor 1.5 measured, profiles/r02_fetch_size_calibration.txt)",
              "avg_launch_ms": round(avg_ms, 5), "flops_per_launch": flops,
              "bytes_per_launch": nbytes, "other_kernel": other,
              "hbm_GBps_algorithmic": round(gbs, 1), "fp32_equivalent_TFLOP/s": round(tf, 3)}
    if b["pipe"] == "hbm":
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), **common}
    if b["pipe"] == "mfma-bf16":
        tfp = pieces * tf
        return {"bound": "mfma", "achieved": round(tfp, 2), "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(tfp / BF16_MFMA_PEAK_TF, 4), **common,
                "achieved_note": f"bf16 flops issued for the algorithmic product: {pieces:g} piece products per fp32 product"}
    return {"bound": "mfma", "achieved": round(tf, 3), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": round(tf / FP32_MFMA_PEAK_TF, 4), **common}


def fit_roofline(n, d, l, n_iter, esz, mode, ms_per_step, steering=0):
    """Whole-fit fraction: the algorithmic floor of one RandomizedPca.fit over the measured time per fit.  The floor is that of
    the cheapest pass structure the product has: n_iter + 1 FUSED passes Y' = Xc^T (Xc P) (one read of X each; 5 + 6 bf16 piece
    products) where the fused kernel exists (bf16x3 mode, d = 512, l <= 80), (2 n_iter + 2) separate GEMM passes otherwise --
    each at its binding roofline -- plus one means pass (n d bytes) and U = Q.Uh (2 n l^2 flop, 2 n l bytes)."""
    fused = mode == "bf16x3" and d == 512 and l <= 80 and n_iter >= 1
    if fused:
        b = pass_bound(4.0 * n * d * l, float(esz) * (n * d + 2 * d * l), mode, 5.5)
        bs = pass_bound(4.0 * n * d * l, float(esz) * (n * d + 2 * d * l), mode, 4.0)   # (a steering pass: four piece products)
        passes = n_iter + 1
        passes_s = (passes - steering) * b["floor_s"] + steering * bs["floor_s"]
    else:
        b = pass_bound(2.0 * n * d * l, float(esz) * (n * d + n * l + d * l), mode)
        passes = 2 * n_iter + 2
        passes_s = passes * b["floor_s"]
    means_s = esz * n * d / (HBM_PEAK_GBS * 1e9)
    u_s = pass_bound(2.0 * n * l * l, float(esz) * 2 * n * l, mode)["floor_s"]
    floor_ms = (passes_s + means_s + u_s) * 1e3
    return {"floor_ms": round(floor_ms, 4), "frac": round(floor_ms / ms_per_step, 4) if ms_per_step > 0 else 0.0,
            "passes": passes, "steering_passes": steering if fused else 0,
            "pass_kind": "fused (one read of X per power iteration)" if fused else "K1 + K2",
            "pass_floor_us": round(b["floor_s"] * 1e6, 2), "pass_pipe": b["pipe"],
            "note": "floor = the passes over X at their binding roofline + the means pass + U = Q.Uh; the serial small-matrix "
                    "steps between the passes have no floor of their own here (serial_chain_ms reports them)"}


def serial_chain(ctx, model, x, omega, ms_per_step, reps=8):
    """The replicated small-matrix chain of a fit: time per fit minus every row-streaming kernel and every all-reduce.  The kernel
    sums come from a few extra fits with EVERY tagged launch bracketed (profiling level 2; the bracketing itself slows those fits,
    so their own wall time is not used): chain = ms_per_step of the timed region - (K1 + K2 + K3 + means pass + U product + comm).
    This is the part of a sharded fit that does not shrink with the number of GPUs."""
    ctx.set_profiling(2)
    sums = []
    try:
        for _ in range(reps):
            model.fit(x, omega=omega)
            st = ctx.stats()
            sums.append({k: st[k] for k in ("xp_ms", "atb_ms", "pow_ms", "stream_ms", "allreduce_ms")} |
                        {"launches": int(st["xp_launches"] + st["atb_launches"] + st["pow_launches"] + st["stream_launches"])})
    finally:
        ctx.set_profiling(True)
    med = {k: float(np.median([s_[k] for s_ in sums])) for k in sums[0]}
    streaming = med["xp_ms"] + med["atb_ms"] + med["pow_ms"] + med["stream_ms"]
    return {"serial_chain_ms": round(ms_per_step - streaming - med["allreduce_ms"], 4),
            "streaming_kernels_ms": round(streaming, 4), "allreduce_ms": round(med["allreduce_ms"], 4),
            "streaming_launches": int(med["launches"]),
            "breakdown_ms": {k: round(v, 4) for k, v in med.items() if k.endswith("_ms")},
            "note": "ms_per_step minus the bracketed row-streaming kernels (K1, K2, fused pass, means pass, U = Z (T Uh)) and "
                    "all-reduces: Cholesky / triangular solve / Gram / eigen-solve / slab combines / launch gaps -- replicated on every rank"}


def predicted_scaling(ms_per_step, chain, n_iter, d, l, strong, allreduce_us=(20.0, 50.0)):
    """What a 2 / 4 / 8-GPU run of THIS workload is expected to give, from figures measured on one GPU: the row-streaming kernels shrink
    with the rank count, the replicated small-matrix chain does not, and a fit adds n_iter + 3 all-reduces (prologue, n_iter + 1 products
    of d x l fp64, the svd_flip key).  Two all-reduce latencies bracket what a <= 1.2 MB RCCL all-reduce over xGMI costs; the first
    measured SCALE line can be read against this.  Strong scaling: one matrix split N ways; weak: N copies of the per-GPU workload."""
    serial = chain["serial_chain_ms"]
    streaming = max(ms_per_step - serial, 0.0)
    calls = n_iter + 3
    res = {"model": "t(N) = streaming_ms / N + serial_chain_ms + (n_iter + 3) x allreduce latency (strong) ; "
                    "t(N) = ms_per_step + (n_iter + 3) x allreduce latency (weak)",
           "streaming_ms": round(streaming, 4), "serial_chain_ms": round(serial, 4), "allreduce_calls_per_fit": calls,
           "allreduce_payload_bytes": 8 * d * l, "assumed_allreduce_us": list(allreduce_us)}
    for N in (2, 4, 8):
        lo_hi = []
        for us in allreduce_us:
            t_strong = streaming / N + serial + calls * us * 1e-3
            t_weak = ms_per_step + calls * us * 1e-3
            lo_hi.append(round(ms_per_step / t_strong, 2) if strong else round(N * ms_per_step / t_weak, 2))
        res[f"speedup_at_{N}_gpus"] = {"best": max(lo_hi), "worst": min(lo_hi)}
    return res


def padded_pitch_extra(petal, ctx, torch, model, x, omega, steps=20, pad_elems=32):
    """The same fit with X held by the CALLER with a padded row pitch (a strided device view, streamed in place): informational.
    A row pitch that is a multiple of 1 KiB puts column chunk c of every row on the same few memory channels; 128 B of padding per
    row spreads them (what the library does by itself for inputs it has to copy)."""
    n, d = x.shape
    buf = torch.empty((n, d + pad_elems), dtype=x.dtype, device=x.device)
    view = buf[:, :d]
    view.copy_(x)
    ramp(lambda: model.fit(view, omega=omega), 0.05)
    t0 = time.perf_counter()
    for _ in range(steps):
        model.fit(view, omega=omega)
    ms = (time.perf_counter() - t0) / steps * 1e3
    st = ctx.stats()
    return {"ms_per_step": round(ms, 4), "samples_per_s": round(n / (ms * 1e-3), 1), "row_pitch_bytes": int(st["x_row_pitch_bytes"]),
            "zero_copy": bool(st["x_zero_copy"]),
            "note": "X passed as a device view with 128 B of padding per row (the caller's layout, streamed in place); `value` above "
                    "is measured on the contiguous matrix"}


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


def exact_mode_extra(petal, ctx, model, x, omega, steps=20):
    """The same fit with PETAL_GEMM_SPLIT_BF16X3_EXACT: three bf16 planes of every operand in every pass (five / six piece products,
    no 16-bit steering passes, no spectrum verdict) -- the split-product arithmetic that is fp32-equivalent throughout."""
    ctx.set_gemm_mode("bf16x3-exact")
    try:
        ramp(lambda: model.fit(x, omega=omega), 0.05)
        t0 = time.perf_counter()
        for _ in range(steps):
            model.fit(x, omega=omega)
        ms = (time.perf_counter() - t0) / steps * 1e3
        st = ctx.stats()
    finally:
        ctx.set_gemm_mode("bf16x3")
    return {"ms_per_step": round(ms, 4), "samples_per_s": round(x.shape[0] / (ms * 1e-3), 1), "rpca_redo": int(st["rpca_redo"]),
            "note": "petal_ctx_set_gemm_mode(PETAL_GEMM_SPLIT_BF16X3_EXACT): every pass on three-plane operands; `value` above is the "
                    "default mode, whose passes before the last run on 16-bit operands behind a spectrum verdict"}


def heavy_tail_matrix(torch, dev, n, d, rho=0.97, seed=11):
    """n x d fp32 with singular values 100 sqrt(n) rho^i over ALL d directions (no planted gap, no noise floor below): the
    slowly decaying spectrum on which the two-plane verdict refuses the optimistic run"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    q, _ = torch.linalg.qr(torch.randn((d, d), generator=g, device=dev, dtype=torch.float32))
    s = 100.0 * float(np.sqrt(n)) * torch.pow(torch.tensor(rho, device=dev, dtype=torch.float32), torch.arange(d, device=dev, dtype=torch.float32))
    x = torch.randn((n, d), generator=g, device=dev, dtype=torch.float32) / float(np.sqrt(n))
    return (x * s) @ q.T


def redo_case_extra(petal, ctx, torch, dev, n, d, k, n_iter, steps=20):
    """A fit that the verdict REDOES: the same shape, singular values 0.97^i (heavy tail).  The optimistic run is thrown away and the
    fit repeated on three-plane operands (petal_stats.rpca_redo = 1): what a user with such data pays in the default mode."""
    x = heavy_tail_matrix(torch, dev, n, d)
    omega = np.random.default_rng(3).standard_normal((d, k + 10)).astype(np.float32)
    m = petal.RandomizedPca(k, ctx=ctx, n_iter=n_iter)
    ramp(lambda: m.fit(x, omega=omega), 0.05)
    redo = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        m.fit(x, omega=omega)
        redo = max(redo, int(ctx.stats()["rpca_redo"]))
    ms = (time.perf_counter() - t0) / steps * 1e3
    del x
    return {"ms_per_step": round(ms, 4), "samples_per_s": round(n / (ms * 1e-3), 1), "rpca_redo": redo,
            "spectrum": "sigma_i = 100 sqrt(n) 0.97^i over all 512 directions",
            "note": "rpca_redo = 1: the spectrum verdict refused the optimistic (16-bit operand) run and the fit was repeated with three "
                    "planes everywhere; PETAL_GEMM_SPLIT_BF16X3_EXACT avoids the first run for callers who know their spectra"}


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
            ramp(fn, 0.1)
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


def northstar_fit(petal, ctx, torch, dev, gemm, n=1_000_000, d=512, k=64, reps=5):
    """The north-star point as a FIT: RandomizedPca.fit on a planted 1e6 x 512 fp32 matrix (2.05 GB, beyond the Infinity Cache),
    k = 64, at 5 power iterations (BASELINE configs[1]'s count) and 7 (the crate's default), with the whole-fit roofline
    fraction: here the serial small-matrix chain is a small share of the time, unlike at 100000 rows."""
    from synth_data import synth_pca_device
    l = k + 10
    x = synth_pca_device(n, d, k, 6, 0, n, dev)
    omega = np.random.default_rng(3).standard_normal((d, l)).astype(np.float32)
    res = {"shape": f"{n}x{d} fp32, k={k}", "gemm_mode": gemm}
    for n_iter in (5, 7):
        m = petal.RandomizedPca(k, ctx=ctx, n_iter=n_iter)
        ramp(lambda: m.fit(x, omega=omega), 0.1)
        torch.cuda.synchronize(dev)
        acc = {"xp_ms": 0.0, "xp_launches": 0, "atb_ms": 0.0, "atb_launches": 0}
        t0 = time.perf_counter()
        for _ in range(reps):
            m.fit(x, omega=omega)
            st = ctx.stats()
            for key in acc:
                acc[key] += st[key]
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / reps * 1e3
        res[f"n_iter_{n_iter}"] = {
            "ms_per_fit": round(ms, 4), "samples_per_s": round(n / (ms * 1e-3), 1),
            "K1_avg_launch_ms": round(acc["xp_ms"] / max(acc["xp_launches"], 1), 4),
            "K2_avg_launch_ms": round(acc["atb_ms"] / max(acc["atb_launches"], 1), 4),
            "fit_roofline": fit_roofline(n, d, l, n_iter, 4, gemm, ms, steering_passes(n, d, l, n_iter, gemm))}
    del x
    torch.cuda.empty_cache()
    return res


def bench_fastica(args, cfg, petal, ctx, torch, dist, dev, rank, world, collective, sync_all):
    """--config cfg5: a step is one FastIca.fit() (whitening + the fixed-point loop to tol 1e-4, src/ica.rs:167-221) on this
    rank's row block; the per-iteration all-reduce carries (nc^2 + nc) fp64 values."""
    from synth_data import synth_ica
    n, d, nc = args.n, args.d, args.k
    x = torch.from_numpy(synth_ica(n, d, nc, seed=8, dtype=np.float32, row_seed=None if world == 1 else 8 + 1000 * rank)).to(dev)
    w0 = np.random.default_rng(7).standard_normal((nc, nc)).astype(np.float32)
    m = petal.FastIca(ctx=ctx, n_components=nc)
    ramp(lambda: m.fit(x, w_init=w0), args.clock_ramp_s, dist, torch, "cpu" if args.share_gpu else dev)   # (clock ramp: see main)
    for _ in range(args.warmup):
        m.fit(x, w_init=w0)
    sync_all()
    step_ms, step_cnt = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.fit(x, w_init=w0)
        st = ctx.stats()
        step_ms += st["ica_step_ms"]
        step_cnt += st["ica_step_launches"]
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if args.share_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    avg = step_ms / max(step_cnt, 1)
    tf = st["ica_step_flops"] / (avg * 1e-3) / 1e12 if avg > 0 else 0.0
    if args.gemm == "fp32":
        roofline = {"bound": "mfma", "achieved": round(tf, 3), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TF, 4),
                    "traffic": None, "kernel": "K7 fused FastICA step (fp32 MFMA)", "avg_launch_ms": round(avg, 5),
                    "flops_per_launch": st["ica_step_flops"], "bytes_per_launch": st["ica_step_bytes"]}
    else:
        # split-product step on pre-split planes (k_ica3p).  Algorithmic work per iteration (SURVEY.md 8 d): 4 nc^2 n flop and 4 nc n
        # bytes -- the fp32 whitened data read once.  Both products form six bf16 piece products per fp32 product, so the candidate
        # floors are 4 nc n / 8 TB/s and 6 x flops / 2.5 PFLOP/s; the kernel's own layout (three bf16 planes of the padded data, 6 B per
        # value) is what it MOVES -- reported as `traffic` (layout bytes, not a PMC reading), never as the floor.
        ncp = -(-nc // 16) * 16
        alg_bytes = 4.0 * nc * n
        plane_bytes = 6.0 * ncp * n
        floors = {"hbm": alg_bytes / (HBM_PEAK_GBS * 1e9), "mfma-bf16": 6.0 * st["ica_step_flops"] / (BF16_MFMA_PEAK_TF * 1e12)}
        pipe = max(floors, key=floors.get)
        common = {"pipe": pipe, "candidate_floors_us": {k: round(v * 1e6, 2) for k, v in floors.items()},
                  "traffic": plane_bytes, "traffic_note": "bytes of the three-plane layout the step kernel streams (6 B per padded value, "
                                                          "1.5 x the algorithmic 4 B): layout arithmetic, not a PMC measurement",
                  "kernel": "K7 fused FastICA step on pre-split planes, k_ica3p (bf16x3 split-product, fp32 accumulate)", "piece_products": 6,
                  "avg_launch_ms": round(avg, 5), "flops_per_launch": st["ica_step_flops"], "bytes_per_launch": alg_bytes,
                  "fp32_equivalent_TFLOP/s": round(tf, 3),
                  "note": "the step kernel is 10 launches of a fit whose largest single launch is the split-product Gram kernel k_gram5 (timelines in profiles/)"}
        if pipe == "hbm":
            gbs = alg_bytes / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
            roofline = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), **common}
        else:
            btf = 6.0 * tf
            roofline = {"bound": "mfma", "achieved": round(btf, 2), "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(btf / BF16_MFMA_PEAK_TF, 4), **common}
    return {
        "metric": "samples/sec for FastIca.fit() on n x d fp32", "value": round(world * n * args.steps / elapsed, 1),
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"FastIca.fit n_components={nc} logcosh tol=1e-4 on {n}x{d} fp32 per GPU ({cfg['label']}), X resident "
                               f"in HBM, {m.n_iter} iterations", "rows_per_gpu": n, "features": d, "n_components": nc,
                   "n_iter": m.n_iter, "parallelism": f"sample-sharded x{world}" if world > 1 else "single GPU",
                   "collective": collective},
        "roofline": roofline,
    }


def pca_cfg1(petal, ctx, torch, dev):
    """BASELINE configs[0] (informational): the exact (full-SVD) Pca on the reference's own CPU-runnable case, 1000 x 16 f64,
    and on a tall fp32 matrix (200000 x 256, k = 32), X in HBM."""
    from synth_data import synth_pca
    res = {}
    for name, (n, d, k, dt) in {"1000x16_f64_k4": (1000, 16, 4, np.float64), "200000x256_f32_k32": (200000, 256, 32, np.float32)}.items():
        x = torch.from_numpy(synth_pca(n, d, k, seed=2, dtype=dt)).to(dev)
        m = petal.Pca.new(k, ctx)
        t_w = time.perf_counter()   # (the host-side data generation above idles the GPU: a quarter of a second of fits lets its clocks
        while time.perf_counter() - t_w < 0.25:   #  ramp up again -- five 1-ms fits did not, and the timed ones then ran 2-5 x slow)
            m.fit(x)
        times = []
        for _ in range(20):
            t0 = time.perf_counter()
            m.fit(x)
            times.append(time.perf_counter() - t0)
        res[name] = {"fit_ms": round(float(np.median(times)) * 1e3, 4), "fit_mean_ms": round(float(np.mean(times)) * 1e3, 4)}
    return res


def fastica_cfg3(petal, ctx, torch, dev, n=200000, d=256, nc=32):
    """BASELINE configs[2] (informational): FastIca n_components=32 (logcosh) on 200000 x 256 fp32, X in HBM.
    Reports the full fit (whitening + loop to the 1e-4 criterion) and the loop at a fixed 200 iterations."""
    from synth_data import synth_ica
    x = torch.from_numpy(synth_ica(n, d, nc, seed=5, dtype=np.float32)).to(dev)
    w0 = np.random.default_rng(7).standard_normal((nc, nc)).astype(np.float32)
    m = petal.FastIca(ctx=ctx, n_components=nc)
    ramp(lambda: m.fit(x, w_init=w0), 0.25)  # the host-side data generation above idles the GPU: let its clocks ramp up again
    reps = 10
    step_sum, step_cnt, times = 0.0, 0, []
    for _ in range(reps):
        t0 = time.perf_counter()
        m.fit(x, w_init=w0)        # (returns after its own synchronisation)
        times.append(time.perf_counter() - t0)
        st = ctx.stats()   # (level-1 profiling samples one launch per fit, the kind rotating: accumulate over the fits)
        step_sum += st["ica_step_ms"]
        step_cnt += st["ica_step_launches"]
    fit_ms = float(np.median(times)) * 1e3   # (median: one host hiccup in ten fits would otherwise own the mean)
    fit_mean_ms = float(np.mean(times)) * 1e3
    step_ms = step_sum / max(step_cnt, 1)
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
    return {"shape": f"{n}x{d} fp32, n_components={nc}", "fit_ms": round(fit_ms, 3), "fit_mean_ms": round(fit_mean_ms, 3), "n_iter": m.n_iter,
            "loop_only_on_whitened_ms": round(loop_ms, 3), "loop_only_n_iter": int(it_loop),
            "samples_per_s": round(n / (fit_ms * 1e-3), 1),
            "fit_fixed_200_iter_ms": round(fit200_ms, 3), "ms_per_iteration": round((fit200_ms - fit_ms) / max(200 - m.n_iter, 1), 4),
            "step_kernel": {"avg_launch_ms": round(step_ms, 5),
                            "TFLOP/s": round(st["ica_step_flops"] / (step_ms * 1e-3) / 1e12, 2) if step_ms > 0 else 0.0,
                            "GB/s_algorithmic": round(st["ica_step_bytes"] / (step_ms * 1e-3) / 1e9, 1) if step_ms > 0 else 0.0}}


def cpu_baseline(x_host, omega, k, n_iter):
    """The oracle (numpy + LAPACK restatement of the reference algorithm, kind "port") on the host cores, same workload,
    same Omega, same n_iter.  Three CPU forms are timed, each with one warm-up fit and the median of three -- (c) is the
    threaded C++ restatement oracle/cpu_rpca.cpp --; of the numpy oracle two thread configurations:
      (a) every BLAS/LAPACK call multithreaded (numpy's OpenBLAS for the `@` products, scipy's for getrf/geqrf/gesdd);
      (b) numpy's OpenBLAS held to ONE thread, scipy's LAPACK on all cores -- the crate's real split: its ndarray GEMMs are
          single-threaded `matrixmultiply` (no blas feature, Cargo.toml:53), only the LAPACK calls reach the threaded backend.
    `value` is the FASTER of the two (the two OpenBLAS thread pools can fight each other in (a)); both are listed."""
    from oracle import petal_oracle as po
    threads = os.cpu_count()
    infos = []
    try:
        from threadpoolctl import threadpool_info
        infos = threadpool_info()
        threads = max([i.get("num_threads", 1) for i in infos] + [1])
    except Exception:
        pass
    m = po.RandomizedPcaOracle(k, n_iter=n_iter)

    def median_of_3():
        m.fit(x_host, omega=omega)  # warm-up: page in LAPACK, spin up the thread pools
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            m.fit(x_host, omega=omega)
            times.append(time.perf_counter() - t0)
        return sorted(times)[1], times

    n = x_host.shape[0]
    dt_a, runs_a = median_of_3()
    variants = {"all_threaded": {"samples_per_s": round(n / dt_a, 1), "median_s": round(dt_a, 3), "runs_s": [round(t, 3) for t in runs_a]}}
    best, how = dt_a, f"GEMMs and LAPACK on {threads} threads"
    try:
        from threadpoolctl import ThreadpoolController
        numpy_blas = [i for i in infos if "numpy.libs" in (i.get("filepath") or "")]
        if numpy_blas and len(infos) > len(numpy_blas):
            ctl = ThreadpoolController().select(filepath=numpy_blas[0]["filepath"])
            with ctl.limit(limits=1):
                dt_b, runs_b = median_of_3()
            variants["single_threaded_gemm"] = {"samples_per_s": round(n / dt_b, 1), "median_s": round(dt_b, 3),
                                                "runs_s": [round(t, 3) for t in runs_b]}
            if dt_b < best:
                best, how = dt_b, f"GEMMs on 1 thread (as the crate's matrixmultiply), LAPACK on {threads} threads"
    except Exception as e:  # informational only
        variants["single_threaded_gemm"] = {"error": repr(e)}
    try:  # (c) the dependency-free threaded C++ restatement (oracle/cpu_rpca.cpp: own GEMM / LU / Householder QR / Jacobi SVD, OpenMP)
        from oracle import cpu_rpca
        mc = cpu_rpca.RandomizedPcaCpp(k, n_iter=n_iter)
        mc.fit(x_host, omega)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            mc.fit(x_host, omega)
            times.append(time.perf_counter() - t0)
        dt_c = sorted(times)[1]
        variants["cpp_restatement"] = {"samples_per_s": round(n / dt_c, 1), "median_s": round(dt_c, 3), "runs_s": [round(t, 3) for t in times],
                                       "threads": cpu_rpca.max_threads()}
        if dt_c < best:
            best, how = dt_c, (f"threaded C++ restatement without BLAS/LAPACK (oracle/cpu_rpca.cpp, OpenMP, {cpu_rpca.max_threads()} threads, "
                               f"fp32 arithmetic like the crate's A = f32)")
            threads = cpu_rpca.max_threads()
    except Exception as e:
        variants["cpp_restatement"] = {"error": repr(e)}
    return {"value": round(n / best, 1), "unit": "samples/s", "cores": int(threads), "kind": "port",
            "sample": f"full fits of the same {n}x{x_host.shape[1]} fp32 workload (k={k}, n_iter={n_iter}), warm-up + median of 3 = "
                      f"{best:.2f} s, the fastest of three CPU forms of the reference algorithm (numpy + OpenBLAS/LAPACK getrf P.L, "
                      f"geqrf/orgqr, gesdd with threaded or single-threaded GEMMs; a threaded C++ restatement): {how}",
            "variants": variants}


if __name__ == "__main__":
    main()

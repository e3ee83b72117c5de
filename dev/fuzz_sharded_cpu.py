"""dev: sample-sharded fits on the host simulation (gloo, 3 ranks) with random row splits -- empty ranks, one-row ranks, uneven
blocks -- against the single-process host-simulation fit of the whole matrix.  CPU only."""
import os, sys, socket, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p
def cases(seed):
    rng = np.random.default_rng(seed)
    out = []
    for c in range(6):
        n = int(rng.choice([40, 301, 1000, 2500])); d = int(rng.choice([5, 16, 33, 64])); k = int(rng.integers(1, max(2, min(n, d) - 1)))
        kind = ["rpca32", "rpca64", "pca64", "ica64"][c % 4]
        w = rng.random(3) ** 3
        if rng.integers(0, 3) == 0: w[rng.integers(0, 3)] = 0.0          # an empty rank
        w = w / max(w.sum(), 1e-9)
        e = np.concatenate([[0], np.round(np.cumsum(w) * n)]).astype(int); e[-1] = n
        out.append((kind, n, d, k, [int(v) for v in e], 900 + 10 * seed + c))
    return out
def variant(sd):
    """per-case flavour of the RandomizedPca cases (round 6: the verdict codes a sharded fit has to agree on): data off centre without
    centering, the eigen-solver's closeness verdict forced (the simulation's hook), rank-deficient data, the iteration count"""
    r = np.random.default_rng(sd + 77)
    return dict(off=float(r.choice([0.0, 0.0, 40.0])), cent=bool(r.integers(0, 2)), hook=bool(r.integers(0, 3) == 0), lowrank=bool(r.integers(0, 4) == 0),
                n_iter=int(r.choice([1, 3, 4, 7])))
def rpca_data(synth_pca, n, d, k, sd, dt, v):
    x = synth_pca(n, d, k, seed=sd, dtype=np.float64)
    if v["lowrank"]:
        r = np.random.default_rng(sd + 5); x = r.standard_normal((n, 2)) @ r.standard_normal((2, d))
    x = x + v["off"] * x.std(axis=0) * np.sign(np.random.default_rng(sd + 6).standard_normal(d))
    return x.astype(dt)
def worker(rank, world, port, out_dir, seed):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import hostsim, petal_decomposition_amd as petal
    from synth_data import synth_ica, synth_pca
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ctx = hostsim.context(); ctx.use_torch_distributed()
    res = {}
    for i, (kind, n, d, k, e, sd) in enumerate(cases(seed)):
        try:
            if kind.startswith("rpca"):
                dt = np.float32 if kind.endswith("32") else np.float64
                v = variant(sd)
                x = rpca_data(synth_pca, n, d, k, sd, dt, v); xs = x[e[rank]:e[rank + 1]]
                om = np.random.default_rng(sd + 1).standard_normal((d, k + 10)).astype(dt)
                ctx.set_option("eigh_jacobi", 2 if v["hook"] else 0)
                m = petal.RandomizedPca(k, centering=v["cent"], ctx=ctx, n_iter=v["n_iter"]); y = m.fit_transform(xs, omega=om)
                ctx.set_option("eigh_jacobi", 0)
                res[f"{i}.comp"], res[f"{i}.sing"], res[f"{i}.y"] = m.components(), m.singular_values(), y
            elif kind == "pca64":
                kk = min(k, 8)
                x = synth_pca(n, d, kk, seed=sd, dtype=np.float64); xs = x[e[rank]:e[rank + 1]]
                p = petal.Pca(kk, ctx=ctx); y = p.fit_transform(xs)
                res[f"{i}.comp"], res[f"{i}.sing"], res[f"{i}.y"] = p.components(), p.singular_values(), y
            else:
                nc = min(d, 6)
                x = synth_ica(max(n, 500), d, nc, seed=sd, dtype=np.float64); ee = [int(round(v * max(n, 500) / n)) for v in e]; ee[-1] = max(n, 500)
                xs = x[ee[rank]:ee[rank + 1]]
                w0 = np.random.default_rng(sd + 2).standard_normal((nc, nc))
                ica = petal.FastIca(ctx=ctx, n_components=nc); y = ica.fit_transform(xs, w_init=w0)
                res[f"{i}.comp"], res[f"{i}.sing"], res[f"{i}.y"] = ica.components, np.array([ica.n_iter], dtype=np.float64), y
        except Exception as ex:
            res[f"{i}.err"] = np.array([str(ex)[:200]])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostsim, petal_decomposition_amd as petal
    from synth_data import synth_ica, synth_pca
    hostsim.build()
    bad = 0
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        with tempfile.TemporaryDirectory() as td:
            mp.spawn(worker, args=(3, free_port(), td, seed), nprocs=3, join=True)
            rs = [np.load(os.path.join(td, f"rank{r}.npz"), allow_pickle=True) for r in range(3)]
            ctx = hostsim.context()
            for i, (kind, n, d, k, e, sd) in enumerate(cases(seed)):
                tag = f"seed {seed} case {i} {kind} n={n} d={d} k={k} cuts={e}"
                if any(f"{i}.err" in r for r in rs):
                    errs = [str(r[f"{i}.err"][0]) if f"{i}.err" in r else "-" for r in rs]
                    # the single-process fit must raise the same kind of error
                    print("ERR ", tag, errs); continue
                if kind.startswith("rpca"):
                    dt = np.float32 if kind.endswith("32") else np.float64
                    v = variant(sd); tag += f" {v}"
                    x = rpca_data(synth_pca, n, d, k, sd, dt, v); om = np.random.default_rng(sd + 1).standard_normal((d, k + 10)).astype(dt)
                    ctx.set_option("eigh_jacobi", 2 if v["hook"] else 0)
                    m = petal.RandomizedPca(k, centering=v["cent"], ctx=ctx, n_iter=v["n_iter"]); y = m.fit_transform(x, omega=om); comp, sing = m.components(), m.singular_values()
                    ctx.set_option("eigh_jacobi", 0)
                    tag += f" redo {ctx.stats()['rpca_redo']} eigh_redo {ctx.stats()['eigh_redo']}"
                    tol = 2e-4 if dt == np.float32 else 1e-8
                elif kind == "pca64":
                    kk = min(k, 8); x = synth_pca(n, d, kk, seed=sd, dtype=np.float64)
                    p = petal.Pca(kk, ctx=ctx); y = p.fit_transform(x); comp, sing = p.components(), p.singular_values(); tol = 1e-8
                else:
                    nc = min(d, 6); x = synth_ica(max(n, 500), d, nc, seed=sd, dtype=np.float64); w0 = np.random.default_rng(sd + 2).standard_normal((nc, nc))
                    ica = petal.FastIca(ctx=ctx, n_components=nc); y = ica.fit_transform(x, w_init=w0); comp, sing = ica.components, np.array([ica.n_iter], dtype=np.float64); tol = 1e-7
                    e = [int(round(v * max(n, 500) / n)) for v in e]; e[-1] = max(n, 500)
                ok = True
                for r in range(3):
                    c_err = np.abs(rs[r][f"{i}.comp"] - comp).max() / max(np.abs(comp).max(), 1e-300)
                    s_err = np.abs(rs[r][f"{i}.sing"] - sing).max() / max(np.abs(sing).max(), 1e-300)
                    yr = rs[r][f"{i}.y"]; yo = np.asarray(y)[e[r]:e[r + 1]]
                    y_err = (np.abs(yr - yo).max() / max(np.abs(y).max(), 1e-300)) if yr.size else 0.0
                    if not (c_err <= tol and s_err <= tol and y_err <= 10 * tol and yr.shape == yo.shape): ok = False; print("   rank", r, "comp", c_err, "sing", s_err, "y", y_err, yr.shape, yo.shape)
                print("ok  " if ok else "FAIL", tag, flush=True)
                bad += 0 if ok else 1
    print("failures:", bad)

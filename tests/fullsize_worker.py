"""One of the EIGHT ranks of BASELINE configs[3] / configs[4] at their stated size (tests/test_gpu_fullsize.py starts them).

    RANK=r WORLD_SIZE=8 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python fullsize_worker.py <out_dir>

Every rank opens its OWN ctx on libpetal_hip.so (on the one-GPU test box all eight share device 0; with eight GPUs visible each
takes its own), generates ITS 250 000 x 1024 / 500 000 x 512 row block of the one planted matrix in device memory
(tests/fullsize_cases.py) and runs the product's sharded fits with the collective hook on a gloo group (the hook stages the
small replicated fp64 buffers through the host: RCCL refuses several ranks on one device).  Results -> <out_dir>/rank<r>.npz.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import petal_decomposition_amd as petal
    import fullsize_cases as fc

    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == fc.WORLD
    local = int(os.environ.get("LOCAL_RANK", str(rank))) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = petal.Context(local)
    ctx.use_torch_distributed()
    res = {}

    # ---- configs[3]: RandomizedPca k = 128, n_iter = 7 (the crate's constant, src/pca.rs:680), 2 000 000 x 1024 over 8 ranks
    c4 = fc.CFG4
    x = fc.cfg4_block(rank)
    m = petal.RandomizedPca(c4["k"], ctx=ctx, n_iter=c4["n_iter"])
    y = m.fit_transform(x, omega=fc.cfg4_omega())
    st = ctx.stats()
    res["cfg4.components"], res["cfg4.singular"] = m.components(), m.singular_values()
    res["cfg4.evr"], res["cfg4.mean"] = m.explained_variance_ratio(), m.mean()
    res["cfg4.allreduce"] = np.array([st["allreduce_calls"], st["allreduce_bytes"]])
    ay = y.abs()
    res["cfg4.colmax"] = ay.max(dim=0).values.cpu().numpy()                      # who owns each column's svd_flip element
    res["cfg4.colarg"] = ay.argmax(dim=0).cpu().numpy() + rank * (c4["n"] // world)
    res["cfg4.xsum"] = np.array([float(x.double().sum())])
    t = m.transform(x[:2000])
    back = m.inverse_transform(t)
    t2 = m.transform(back)
    res["cfg4.idem"] = np.array([float((t2 - t).abs().max()), float(t.abs().max())])  # transform . inverse_transform is a projection
    res["cfg4.y_vs_transform"] = np.array([float((t - y[:2000]).abs().max()), float(y[:2000].abs().max())])
    del x, y, ay, t, back, t2
    torch.cuda.empty_cache()

    # ---- configs[4]: FastIca, 64 components, tol 1e-4, 4 000 000 x 512 over 8 ranks
    c5 = fc.CFG5
    x, src = fc.cfg5_block(rank, want_sources=True)
    ica = petal.FastIca(ctx=ctx, n_components=c5["nc"], tol=c5["tol"])
    y = ica.fit_transform(x, w_init=fc.cfg5_w0())
    st = ctx.stats()
    res["cfg5.components"], res["cfg5.mean"] = ica.components, ica.means
    res["cfg5.n_iter"] = np.array([ica.n_iter])
    res["cfg5.allreduce"] = np.array([st["allreduce_calls"], st["allreduce_bytes"]])
    res["cfg5.corr"] = fc.source_match(y, src)                                   # this rank's samples: sources recovered?
    res["cfg5.xsum"] = np.array([float(x.double().sum())])
    del x, y, src
    ctx.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Stand-alone check of the library's built-in RCCL path (petal_ctx_init_rccl): run as one child process per rank,

    RANK=r WORLD_SIZE=w LOCAL_RANK=g MASTER_ADDR=127.0.0.1 MASTER_PORT=p python rccl_probe.py

it creates the communicator, runs one small sample-sharded RandomizedPca.fit (which all-reduces the means, the l x l Gram
blocks and Xc^T Z on the library's stream) and verifies that every rank ends with the same components.  Exit code 0 = the
built-in collective works on this node.  A launcher (bench.py) runs it under a timeout BEFORE creating the communicator in
its own process: a collective that cannot complete on some node then costs a killed child, not a hung job.
The only use of torch.distributed here is a gloo group that carries the 128-byte ncclUniqueId and the final comparison.
"""
import os
import sys

import numpy as np


def main() -> int:
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # the repo root's import shim
    import petal_decomposition_amd as petal

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    ctx = petal.Context(local)
    if world == 1:
        ctx.set_option("force_collective", 1)   # (a one-rank communicator still takes the sharded code path)
    ctx.use_rccl()
    info = ctx.collective_info()
    # the communicator itself must say it spans the job (a library without the query reports -1: not a failure)
    if info["kind"] != "rccl" or info["ncclCommCount"] not in (-1, world) or info["ncclCommUserRank"] not in (-1, rank):
        print(f"rccl_probe: rank {rank}: communicator reports {info}, expected {world} ranks", file=sys.stderr)
        return 1
    n, d, k = 256, 32, 4
    rng = np.random.default_rng(100 + rank)
    x = (rng.standard_normal((n, k)) @ np.random.default_rng(5).standard_normal((k, d))
         + 0.01 * rng.standard_normal((n, d))).astype(np.float32)
    omega = np.random.default_rng(6).standard_normal((d, k + 4)).astype(np.float32)
    model = petal.RandomizedPca(k, ctx=ctx, n_iter=2, n_oversample=4)
    model.fit(x, omega=omega)
    if ctx.stats()["allreduce_calls"] < 2 + 3:    # prologue, n_iter + 1 products, the svd_flip key: the collective must have RUN
        print(f"rccl_probe: rank {rank}: the fit did not go through the collective: {ctx.stats()}", file=sys.stderr)
        return 1
    comp = torch.from_numpy(np.ascontiguousarray(model.components(), dtype=np.float64))
    ref = comp.clone()
    dist.broadcast(ref, src=0)
    same = bool(torch.equal(ref, comp)) and bool(torch.isfinite(comp).all())
    flag = torch.tensor([1 if same else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    return 0 if int(flag.item()) == 1 else 1


if __name__ == "__main__":
    sys.exit(main())

"""One rank of the sample-sharded HIP path (tests/test_gpu_sharded.py starts WORLD_SIZE of these as child processes).

    RANK=r WORLD_SIZE=w MASTER_ADDR=127.0.0.1 MASTER_PORT=p python sharded_gpu_worker.py <out_dir> [backend]

Every rank opens its OWN ctx on GPU (LOCAL_RANK mod visible devices) -- on the one-GPU test box all ranks share device 0
-- holds an uneven row block of each test matrix in DEVICE memory and runs the product's fits through libpetal_hip.so with
the collective hook on a torch.distributed group (gloo: the hook stages the small fp64 buffers through the host, RCCL
refuses two ranks on one device; nccl when every rank has its own GPU).  Results go to <out_dir>/rank<r>.npz.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cuts(n, world):
    """uneven row blocks: rank 0 gets the biggest share"""
    w = np.arange(world, 0, -1, dtype=np.float64) + 0.37
    edges = np.concatenate([[0], np.round(np.cumsum(w) / w.sum() * n)]).astype(int)
    edges[-1] = n
    return edges


def main():
    import torch
    import torch.distributed as dist
    import petal_decomposition_amd as petal
    import sharded_cases as sc

    out_dir = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else "gloo"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", str(rank))) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for mode in ("bf16x3", "fp32"):
        ctx = petal.Context(local)
        ctx.set_gemm_mode(mode)
        ctx.use_torch_distributed()
        for name, case in sc.CASES.items():
            if mode == "fp32" and not case.get("both_modes", False):
                continue
            x = case["x"]()
            e = cuts(x.shape[0], world)
            xs = torch.from_numpy(np.ascontiguousarray(x[e[rank]:e[rank + 1]])).cuda()
            for key, val in case["run"](petal, ctx, xs, rank).items():
                res[f"{name}.{mode}.{key}"] = np.asarray(val.cpu() if hasattr(val, "cpu") else val)
        ctx.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

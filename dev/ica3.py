import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import petal_decomposition_amd as petal
import bench
ctx = petal.Context(0, stream=torch.cuda.current_stream().cuda_stream); ctx.set_profiling(2)
print(json.dumps(bench.fastica_cfg3(petal, ctx, torch, torch.device("cuda", 0))))

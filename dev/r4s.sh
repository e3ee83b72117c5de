#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash dev/final_round4.sh > gpurun_out/final4.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final4_bench_driver_style.json
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/final_gpu_suite.log | tail -3

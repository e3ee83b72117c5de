#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 120 python bench.py --steps 3 --warmup 1 --no-northstar --gpus 2 --share-gpu --no-cpu-baseline > gpurun_out/r6_ab_out.txt 2> gpurun_out/r6_ab_err.txt; echo "rep $rep rc=$?"
done
timeout 120 python bench.py --config cfg5 --steps 2 --warmup 1 --gpus 2 --share-gpu --no-cpu-baseline 2>/dev/null | tail -c 200; echo " cfg5 rc=$?"
timeout 3400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_ab_gpu_suite.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6_ab_gpu_suite.txt | tail -3

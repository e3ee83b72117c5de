#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 3400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_as_gpu_suite.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6_as_gpu_suite.txt | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r6_as_bench_driver.json; python -c "
import json
d=json.load(open('gpurun_out/r6_as_bench_driver.json')); print('driver-style', d['ms_per_step'], d['serial_chain']['serial_chain_ms'], d['roofline']['frac'])"

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_verdict.py -m gpu -s -q > gpurun_out/r6_am_verdict.txt 2>&1; tail -3 gpurun_out/r6_am_verdict.txt | cut -c1-200
timeout 3400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_am_gpu_suite.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6_am_gpu_suite.txt | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r6_am_bench_driver.json
python -c "
import json
d=json.load(open('gpurun_out/r6_am_bench_driver.json')); print('driver-style', d['ms_per_step'], d['value'], d['serial_chain']['serial_chain_ms'], d['roofline']['frac'])"

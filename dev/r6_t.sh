#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python dev/r6_case_a.py 2>&1 | grep "n_iter" | cut -c1-420
FUZZ6_ONLY=off timeout 1500 python dev/fuzz_round6.py 71 40 > gpurun_out/r6_t_off.txt 2>&1; grep "FAIL\|failures" gpurun_out/r6_t_off.txt | cut -c1-260
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_verdict.py tests/test_gpu_eigh.py -x -q -m gpu > gpurun_out/r6_t_parity.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6_t_parity.txt | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/r6_t_bench.json
python -c "
import json
d=json.load(open('gpurun_out/r6_t_bench.json')); print('cfg2', d['ms_per_step'], d['serial_chain']['serial_chain_ms'])"

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "trajectory or uncentred or close_eigen" 2>&1 | tail -2
bash dev/final_round6.sh > gpurun_out/final6_log.txt 2>&1; tail -2 gpurun_out/final6_log.txt
for f in default driver cfg4 cfg5 fp32 cfg4s_1gpu cfg4s_share2; do python -c "
import json,sys
d=json.load(open('gpurun_out/final6_bench_$f.json')); print('$f', d.get('ms_per_step'), d.get('value'), (d.get('serial_chain') or {}).get('serial_chain_ms'), (d.get('roofline') or {}).get('frac'))" 2>&1 | tail -1; done
grep -c "^ok" gpurun_out/final6_fuzz_round6.txt gpurun_out/final6_fuzz_round6_fp32.txt; grep "FAIL\|failures" gpurun_out/final6_fuzz_round6.txt gpurun_out/final6_fuzz_round6_fp32.txt | cut -c1-250
tail -3 gpurun_out/final6_soak.txt; grep -c " ok" gpurun_out/final6_eig_stress.txt

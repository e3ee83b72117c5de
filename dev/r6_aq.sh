#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for s in 101 102; do timeout 1500 python dev/fuzz_round6.py $s 40 > gpurun_out/r6_aq_$s.txt 2>&1; echo "seed $s: $(grep -c '^ok' gpurun_out/r6_aq_$s.txt) ok"; grep "FAIL\|failures" gpurun_out/r6_aq_$s.txt | cut -c1-250; done
FUZZ_GEMM=fp32 timeout 1500 python dev/fuzz_round6.py 103 40 > gpurun_out/r6_aq_103.txt 2>&1; echo "seed 103 fp32: $(grep -c '^ok' gpurun_out/r6_aq_103.txt) ok"; grep "FAIL\|failures" gpurun_out/r6_aq_103.txt | cut -c1-250

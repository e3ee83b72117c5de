#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for s in 111 112 113; do FUZZ6_ONLY=wide timeout 900 python dev/fuzz_round6.py $s 60 > gpurun_out/r6_av_$s.txt 2>&1; echo "seed $s: $(grep -c '^ok' gpurun_out/r6_av_$s.txt) ok, n_iter 5/7: $(grep -c 'n_iter=[57]' gpurun_out/r6_av_$s.txt)"; grep "FAIL" gpurun_out/r6_av_$s.txt | cut -c1-230; done
FUZZ6_ONLY=wide FUZZ_GEMM=fp32 timeout 900 python dev/fuzz_round6.py 114 60 > gpurun_out/r6_av_114.txt 2>&1; echo "seed 114 fp32: $(grep -c '^ok' gpurun_out/r6_av_114.txt) ok, n_iter 5/7: $(grep -c 'n_iter=[57]' gpurun_out/r6_av_114.txt)"; grep "FAIL" gpurun_out/r6_av_114.txt | cut -c1-230

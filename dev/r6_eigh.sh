#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_eigh.py tests/test_gpu_verdict.py -x -q -m gpu -s > gpurun_out/r6_$1_eigh.txt 2>&1; grep -E "passed|failed|error|false negatives|FALSE" gpurun_out/r6_$1_eigh.txt | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "means_folded or rpca or steering or ica or pca" > gpurun_out/r6_$1_parity.txt 2>&1; tail -2 gpurun_out/r6_$1_parity.txt
bash dev/tl.sh r6_$1_rp2 dev/rpca_one.py > /dev/null 2>&1
bash dev/tl.sh r6_$1_rp4 dev/rpca_one.py cfg4 > /dev/null 2>&1
grep -E "tridiag|trieig" gpurun_out/tl_r6_$1_rp2.txt | head -4; grep -E "tridiag|trieig" gpurun_out/tl_r6_$1_rp4.txt | head -4
head -1 gpurun_out/tl_r6_$1_rp2.txt; head -1 gpurun_out/tl_r6_$1_rp4.txt

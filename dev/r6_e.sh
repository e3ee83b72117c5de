#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python dev/ica_accept_probe.py > gpurun_out/r6_e_ica_probe.txt 2>&1; grep gauss gpurun_out/r6_e_ica_probe.txt
timeout 900 python -m pytest tests/test_gpu_verdict.py -x -q -m gpu -s > gpurun_out/r6_e_verdict.txt 2>&1; grep -E "passed|failed|false negatives|FALSE|REDONE" gpurun_out/r6_e_verdict.txt | tail -8
bash dev/r6_check.sh e
grep -E "chol|trsm" gpurun_out/tl_r6_e_rp2.txt | tail -6

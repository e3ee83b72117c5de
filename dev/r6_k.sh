#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "ica or pca or Pca or fullsize or full_size or exact" > gpurun_out/r6_k_parity.txt 2>&1; tail -2 gpurun_out/r6_k_parity.txt
python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_k_cfg5.json
python -c "
import json; d=json.load(open('gpurun_out/r6_k_cfg5.json')); print('cfg5', d['ms_per_step'])"
bash dev/tl.sh r6_k_ica3 dev/ica_one.py > /dev/null 2>&1; head -1 gpurun_out/tl_r6_k_ica3.txt; grep -E "chol|trsm" gpurun_out/tl_r6_k_ica3.txt | tail -4
python dev/fuzz_all.py 21 40 ica 2>&1 | grep -E "FAIL|failures" | head; python dev/fuzz_all.py 22 40 pca 2>&1 | grep -E "FAIL|failures" | head

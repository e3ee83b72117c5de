#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rebased or close_eigen or three_iter or without_a_fact" > gpurun_out/r6_ak.txt 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r6_ak.txt | head -5
python dev/r6_case_a.py 2>&1 | grep "n_iter" | cut -c1-460

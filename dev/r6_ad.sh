#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python dev/r6_case_b.py 2>&1 | grep "off=40" | cut -c1-330
FUZZ6_ONLY=pca timeout 900 python dev/fuzz_round6.py 81 60 2>&1 | grep "FAIL\|failures" | cut -c1-240
FUZZ6_ONLY=pca timeout 900 python dev/fuzz_round6.py 82 60 2>&1 | grep "FAIL\|failures" | cut -c1-240
FUZZ6_ONLY=ica timeout 900 python dev/fuzz_round6.py 83 40 2>&1 | grep "FAIL\|failures" | cut -c1-240
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_eigh.py -x -q -m gpu -k "pca or ica or eigh" 2>&1 | tail -2

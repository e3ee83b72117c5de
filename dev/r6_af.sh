#!/bin/bash
cd "$GRAFT_REPO_ROOT"
FUZZ6_ONLY=ica timeout 900 python dev/fuzz_round6.py 83 40 2>&1 | grep "FAIL\|failures" | cut -c1-240
FUZZ6_ONLY=ica timeout 900 python dev/fuzz_round6.py 84 40 2>&1 | grep "FAIL\|failures" | cut -c1-240
FUZZ6_ONLY=ica FUZZ_GEMM=fp32 timeout 900 python dev/fuzz_round6.py 85 30 2>&1 | grep "FAIL\|failures" | cut -c1-240

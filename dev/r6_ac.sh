#!/bin/bash
cd "$GRAFT_REPO_ROOT"
FUZZ6_ONLY=pca timeout 900 python dev/fuzz_round6.py 81 60 > gpurun_out/r6_ac_pca.txt 2>&1; grep "FAIL\|failures" gpurun_out/r6_ac_pca.txt | cut -c1-240

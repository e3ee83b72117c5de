#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for tree in dev/r5tree .; do
  echo "=== tree $tree"
  (cd $tree && FUZZ6_ONLY=wide timeout 900 python dev/fuzz_round6.py 61 40 2>&1 | grep "FAIL\|failures" | cut -c1-230)
  (cd $tree && FUZZ6_ONLY=wide FUZZ_GEMM=fp32 timeout 900 python dev/fuzz_round6.py 62 25 2>&1 | grep "FAIL\|failures" | cut -c1-230)
done

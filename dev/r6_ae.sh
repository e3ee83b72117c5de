#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for tree in dev/headtree .; do
  echo "=== tree $tree"
  (cd $tree && FUZZ6_ONLY=ica timeout 900 python dev/fuzz_round6.py 83 40 2>&1 | grep "FAIL\|failures\|n=20000 d=128 nc=8" | cut -c1-240)
done

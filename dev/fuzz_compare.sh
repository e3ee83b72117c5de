#!/bin/bash
# the same fuzz seeds on the round-4 tree (dev/r4tree, a git worktree of 0caef4c) and on this tree: FAIL lines side by side
cd "$GRAFT_REPO_ROOT"
for tree in dev/r4tree .; do
  echo "=== tree $tree"
  (cd $tree && timeout 900 python dev/fuzz_all.py 11 60 2>&1 | grep "FAIL" | cut -c1-150)
  (cd $tree && timeout 600 python dev/fuzz_rpca.py 13 60 2>&1 | grep "FAIL\|failures" | cut -c1-150)
done

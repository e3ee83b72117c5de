#!/bin/bash
cd "$GRAFT_REPO_ROOT"
PETAL_HIP_LIBRARY=$PWD/dev/libpetal_orthold.so timeout 900 python dev/fuzz_all.py 21 40 ica 2>&1 | grep -E "FAIL|failures|nc=48" | cut -c1-200
echo "--- new"
timeout 900 python dev/fuzz_all.py 21 40 ica 2>&1 | grep -E "FAIL|failures|nc=48" | cut -c1-200

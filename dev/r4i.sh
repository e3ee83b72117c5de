#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== profiling levels"; python dev/prof_levels.py 2>/dev/null
echo "== full GPU suite"; timeout 3300 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5

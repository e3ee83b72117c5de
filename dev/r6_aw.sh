#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 3400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_aw_gpu_suite.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6_aw_gpu_suite.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1

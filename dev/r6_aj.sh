#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r6_aj.txt 2>&1; grep -n "passed\|failed\|Error\|assert" gpurun_out/r6_aj.txt | head -12

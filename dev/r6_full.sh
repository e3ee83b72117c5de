#!/bin/bash
# round 6: the whole GPU suite + the verdict table + the fuzz classification (outputs under gpurun_out/r6_<tag>_*)
tag=${1:-full}
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu -s > gpurun_out/r6_${tag}_gputests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r6_${tag}_gputests.txt | tail -3
grep -E "false negatives|FALSE" gpurun_out/r6_${tag}_gputests.txt | tail -5
timeout 600 python dev/fuzz_classify.py 13 60 > gpurun_out/r6_${tag}_fuzz_classify.txt 2>&1; tail -15 gpurun_out/r6_${tag}_fuzz_classify.txt

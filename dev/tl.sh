#!/bin/bash
# usage: dev/tl.sh <tag> <script> [args...] -- rocprofv3 kernel trace + timeline of the last fit
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -- python3 "$@" > gpurun_out/tl_$tag.log 2>&1
grep -v simple_timer gpurun_out/tl_$tag.log | tail -4
python3 dev/timeline.py gpurun_out/tl_$tag 300 1 > gpurun_out/tl_$tag.txt
rm -rf gpurun_out/tl_$tag

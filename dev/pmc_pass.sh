#!/bin/bash
# usage: [PMC_SCRIPT=dev/pmc_ica.py] dev/pmc_pass.sh <tag> <rows> <counter> [<counter> ...]   -- one rocprofv3 --pmc pass over dev/pmc_kernels.py
# (counters in their own run, --kernel-trace only; output under gpurun_out/pmc_<tag>)
tag=$1; rows=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 ${PMC_SCRIPT:-dev/pmc_kernels.py} $rows > gpurun_out/pmc_$tag.log 2>&1
tail -1 gpurun_out/pmc_$tag.log

#!/bin/bash
# usage: dev/kt.sh <tag> <kernel-name-pattern> <script> [args...]  -- rocprofv3 kernel trace of a script, averages of matching kernels
tag=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$tag -- python3 "$@" > gpurun_out/kt_$tag.log 2>&1
grep -v simple_timer gpurun_out/kt_$tag.log | tail -2
python3 - "$tag" "$pat" <<'PY'
import csv, glob, sys, re
f = glob.glob(f"gpurun_out/kt_{sys.argv[1]}/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r["Name"]):
        print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:10.1f} us")
PY

#!/bin/bash
# usage: dev/prof_fit.sh <tag> [bench args]  -- rocprofv3 kernel trace of a short cfg2 bench, per-kernel averages
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$tag -- python3 bench.py --steps 30 --warmup 5 --no-northstar --no-cpu-baseline "$@" > gpurun_out/kt_$tag.log 2>&1
tail -1 gpurun_out/kt_$tag.log | cut -c1-200
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/kt_{sys.argv[1]}/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
fits = 38.0  # 30 timed + 5 warm-up + 3 host_in... (approximate: per-fit = total / calls-per-fit)
tot = 0
for r in rows:
    calls = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e3
    tot += calls * avg
    print(f"{r['Name'][:86]:86s} {calls:6d} {avg:9.1f} us  {calls*avg/1e3:9.2f} ms")
print("total kernel ms", tot / 1e3)
PY

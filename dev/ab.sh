#!/bin/bash
# usage: dev/ab.sh LIB_A LIB_B [script args...] -- A/B of two libraries on the SAME box: alternating bench runs (configs[1] fit) and one kernel-stats pass each
cd "$GRAFT_REPO_ROOT"
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in $A $B; do
    PETAL_HIP_LIBRARY=$lib python bench.py --no-cpu-baseline --no-northstar --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$lib', r['ms_per_step'], r['roofline']['avg_launch_ms'], list(r['roofline']['other_kernel'].values())[0]['avg_launch_ms'])"
  done
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -q -m gpu -x -k "rpca or two_plane or sharded or determinism or config" 2>&1 | tail -4
for knob in "" "PETAL_NO_POW3_FAST=1"; do
  echo "== $knob"
  env $knob python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('cfg4', r['ms_per_step'], r['roofline'].get('avg_launch_ms'), r.get('rpca_redo'))"
done

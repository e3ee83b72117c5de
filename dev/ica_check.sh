#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -q -m gpu -k "ica or gram or whiten or Ica" 2>&1 | tail -5
python dev/ica_bench.py 2>&1 | grep -v amdgpu.ids | tail -3
python dev/ica5_bench.py 2>&1 | grep -v amdgpu.ids | tail -3
python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('cfg5', r['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('cfg2', r['ms_per_step'], r.get('fastica_cfg3'))"

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu -k "without_a_factorisation or sharded or rank" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('cfg2', d['ms_per_step'])"

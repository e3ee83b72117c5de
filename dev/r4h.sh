#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== eigh tests (look-ahead kernel)"; timeout 900 python -m pytest tests/test_gpu_eigh.py -x -q -m gpu > gpurun_out/r4h_eigh.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r4h_eigh.log | tail -5
for rep in 1 2; do
echo "== tridiag new"; bash dev/kt.sh r4h_new "tridiag|trieig" dev/chol_bench.py
echo "== tridiag old"; PETAL_TRIDIAG_OLD=1 bash dev/kt.sh r4h_old "tridiag|trieig" dev/chol_bench.py
done
echo "== fits new/old alternating"
for rep in 1 2 3; do
python bench.py --no-cpu-baseline --no-northstar --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('new', r['ms_per_step'])"
PETAL_TRIDIAG_OLD=1 python bench.py --no-cpu-baseline --no-northstar --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('old', r['ms_per_step'])"
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== fastica timeline (side-stream decorrelation)"; bash dev/tl.sh r4k_ica dev/ica_one.py; grep "^#" gpurun_out/tl_r4k_ica.txt | head -4; head -20 gpurun_out/tl_r4k_ica.txt
echo "== ica tests"; timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -m gpu -x -q -k "ica or Ica or cfg5 or kats or determinism or edge or replicated or single_process" 2>&1 | grep -E "passed|failed|Error" | tail -3
echo "== poison"; PETAL_POISON=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "uninitialised or ica_parity" 2>&1 | grep -E "passed|failed|Error" | tail -3
echo "== bench fastica"; python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['fastica_cfg3'])"
python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('cfg5', r['ms_per_step'])"

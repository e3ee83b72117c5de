#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in cfg2 cfg4 cfg5; do
  timeout 200 python bench.py --config $cfg --steps 3 --warmup 1 --no-northstar --gpus 2 --share-gpu --no-cpu-baseline 2> gpurun_out/r6_at_$cfg.err | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$cfg', 'n_gpus', d['n_gpus'], 'scaling', d['scaling'], 'ms', d['ms_per_step'], 'chain', (d.get('serial_chain') or {}).get('serial_chain_ms'), 'speedup key', 'speedup_vs_one_gpu_same_matrix' in d)
except Exception as e: print('$cfg FAILED', e)"
done
timeout 200 python bench.py --steps 3 --warmup 1 --no-northstar --gpus 2 --share-gpu --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default N=2', d['scaling'], d['ms_per_step'], d['speedup_vs_one_gpu_same_matrix'])"

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== bench"; python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4o_bench.json; python -c "
import json; r=json.load(open('gpurun_out/r4o_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['fastica_cfg3']['fit_ms'], r['pca_cfg1'], r['northstar_fit']['n_iter_5']['ms_per_fit'])"
echo "== timeline rpca tail"; bash dev/tl.sh r4o_rp2 dev/rpca_one.py; tail -28 gpurun_out/tl_r4o_rp2.txt | head -8
echo "== pca timeline"; bash dev/tl.sh r4o_pca dev/pca_one.py; grep "^#" gpurun_out/tl_r4o_pca.txt | head -12
echo "== full suite"; timeout 3300 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3

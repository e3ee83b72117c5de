#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== fastica timeline"; bash dev/tl.sh r4m_ica dev/ica_one.py; grep "^# " gpurun_out/tl_r4m_ica.txt | head -3; sed -n 2,40p gpurun_out/tl_r4m_ica.txt
echo "== A/B nn gemm: fits"; python dev/ica_bench.py 2>/dev/null | tail -3; PETAL_NO_GEMM_NN=1 python dev/ica_bench.py 2>/dev/null | tail -3
echo "== tests"; timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
echo "== bench"; python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4m_bench.json; python -c "
import json; r=json.load(open('gpurun_out/r4m_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['fastica_cfg3']['fit_ms'], r['pca_cfg1'], r['northstar_fit']['n_iter_5']['ms_per_fit'])"

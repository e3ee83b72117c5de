#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== bench default as the driver runs it"; python3 bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r4j_bench.err | tail -1 > gpurun_out/r4j_bench.json; python -c "
import json; r=json.load(open('gpurun_out/r4j_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['fastica_cfg3']['fit_ms'], r['fastica_cfg3']['fit_mean_ms'], r['fastica_cfg3']['ms_per_iteration'], r['pca_cfg1'], r['x_padded_pitch']['ms_per_step'], r['northstar_fit']['n_iter_5']['ms_per_fit'])"
echo "== without ramp"; python3 bench.py --gpus 1 --steps 20 --warmup 5 --clock-ramp-s 0 --no-northstar --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'])"
echo "== with ramp again"; python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-northstar --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'])"
echo "== tests"; timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3

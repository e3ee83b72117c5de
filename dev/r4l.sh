#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== rpca timeline"; bash dev/tl.sh r4l_rp2 dev/rpca_one.py; head -9 gpurun_out/tl_r4l_rp2.txt; tail -22 gpurun_out/tl_r4l_rp2.txt | head -8
echo "== fastica timeline"; bash dev/tl.sh r4l_ica dev/ica_one.py; head -9 gpurun_out/tl_r4l_ica.txt
echo "== bench as the driver"; python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4l_bench.json; python -c "
import json; r=json.load(open('gpurun_out/r4l_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['fastica_cfg3']['fit_ms'], r['pca_cfg1'], r['northstar_fit']['n_iter_5']['ms_per_fit'])"
python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('cfg5', r['ms_per_step'])"
python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('cfg4', r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['other_kernel'])"
echo "== full suite"; timeout 3300 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
echo "== poison"; PETAL_POISON=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "uninitialised or rpca_parity or ica_parity or pca_parity" 2>&1 | grep -E "passed|failed|Error" | tail -3

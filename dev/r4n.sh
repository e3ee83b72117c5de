#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== K2 wide: exactness"; PETAL_HIP_LIBRARY=dev/libpetal_k2wide.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exact or rpca_parity" 2>&1 | grep -E "passed|failed|Error" | tail -3
echo "== A/B: default vs K2 wide"; bash dev/ab.sh petal-decomposition_amd/libpetal_hip.so dev/libpetal_k2wide.so
echo "== cfg4 A/B"; for lib in petal-decomposition_amd/libpetal_hip.so dev/libpetal_k2wide.so petal-decomposition_amd/libpetal_hip.so dev/libpetal_k2wide.so; do PETAL_HIP_LIBRARY=$lib python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$lib', r['ms_per_step'], r['roofline']['avg_launch_ms'], list(r['roofline']['other_kernel'].values())[0]['avg_launch_ms'])"; done
echo "== colsum4: timeline + tests"; bash dev/tl.sh r4n_rp4 dev/rpca_one.py cfg4; grep "colsum" gpurun_out/tl_r4n_rp4.txt | head -3; bash dev/tl.sh r4n_rp2 dev/rpca_one.py; grep "colsum" gpurun_out/tl_r4n_rp2.txt | head -2
timeout 2500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3

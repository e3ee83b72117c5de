#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== A/B cholesky chain: LDLt (default lib) vs Cholesky form"
bash dev/ab.sh petal-decomposition_amd/libpetal_hip.so dev/libpetal_cholform.so
echo "== kernel stats LDLt"; bash dev/kt.sh r4h_a "k_chol_inv2" dev/chol_bench.py
echo "== kernel stats chol form"; PETAL_HIP_LIBRARY=dev/libpetal_cholform.so bash dev/kt.sh r4h_b "k_chol_inv2" dev/chol_bench.py
echo "== fullsize + sharded tests"; timeout 3000 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_eigh.py tests/test_cpp_facade.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
rm -rf gpurun_out/kt_r4h_*

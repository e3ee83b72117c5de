#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== chol LDLt"; bash dev/kt.sh r4g_chol "k_chol_inv2|k_trsm" dev/chol_bench.py
echo "== full GPU suite"; timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
echo "== bench"; python bench.py --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/r4g_bench.json; python -c "
import json; r=json.load(open('gpurun_out/r4g_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'])"
rm -rf gpurun_out/kt_r4g_chol

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 3400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_ap_gpu_suite.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6_ap_gpu_suite.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('driver-style', d['ms_per_step'], d['serial_chain']['serial_chain_ms'], d['roofline']['frac'], d['exact_mode']['ms_per_step'], d['redo_case']['ms_per_step'], d['fp32_mfma_mode']['ms_per_step'])"

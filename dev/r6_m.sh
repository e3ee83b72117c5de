#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/r6_m_bench.json
python -c "
import json
d=json.load(open('gpurun_out/r6_m_bench.json')); print('cfg2', d['ms_per_step'], d['serial_chain']['serial_chain_ms'])"
bash dev/tl.sh r6_m_rp2 dev/rpca_one.py > /dev/null 2>&1
grep -E "chol|trsm" gpurun_out/tl_r6_m_rp2.txt | head -12; head -1 gpurun_out/tl_r6_m_rp2.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rpca or fused or steer or pow or golden or fullsize" > gpurun_out/r6_m_parity.txt 2>&1; tail -2 gpurun_out/r6_m_parity.txt

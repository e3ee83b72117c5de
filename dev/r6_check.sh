#!/bin/bash
# round 6: parity subset + bench line + timelines after a kernel change (outputs under gpurun_out/r6_<tag>_*)
tag=${1:-a}
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r6_${tag}_parity.txt 2>&1; tail -3 gpurun_out/r6_${tag}_parity.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/r6_${tag}_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r6_${tag}_bench.json')); print('ms_per_step', d['ms_per_step'], 'roofline', d['roofline'], 'chain', d.get('serial_chain'))"
bash dev/tl.sh r6_${tag}_rp2 dev/rpca_one.py > /dev/null 2>&1
bash dev/tl.sh r6_${tag}_rp4 dev/rpca_one.py cfg4 > /dev/null 2>&1
grep -E "^#   " gpurun_out/tl_r6_${tag}_rp2.txt | head -12
grep -E "^#   " gpurun_out/tl_r6_${tag}_rp4.txt | head -12
if [ "$2" = "dbg" ]; then
  bash dev/build_dbg.sh > /dev/null 2>&1
  python dev/dbg_phases.py > gpurun_out/r6_${tag}_dbg_phases.txt 2>&1; tail -3 gpurun_out/r6_${tag}_dbg_phases.txt
fi

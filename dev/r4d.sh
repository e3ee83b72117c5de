#!/bin/bash
# round 4, batch d: the ticketed re-basing kernel with sc1 hand-offs -- timelines, bench, parity + determinism
cd "$GRAFT_REPO_ROOT"
echo "== timeline cfg2 fused"; bash dev/tl.sh r4d_rp2 dev/rpca_one.py; tail -24 gpurun_out/tl_r4d_rp2.txt
echo "== timeline cfg4 fused"; bash dev/tl.sh r4d_rp4 dev/rpca_one.py cfg4; grep "^#" gpurun_out/tl_r4d_rp4.txt | head -14
echo "== bench"; python bench.py --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/r4d_bench.json; python -c "
import json; r=json.load(open('gpurun_out/r4d_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'])"
echo "== parity"; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4

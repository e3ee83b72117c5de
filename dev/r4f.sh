#!/bin/bash
# round 4, batch f: two-plane P (K1 with five piece products) -- parity, timelines, bench, A/B against PETAL_NO_P2
cd "$GRAFT_REPO_ROOT"
echo "== timeline cfg2 P2"; bash dev/tl.sh r4f_rp2 dev/rpca_one.py; grep "^#" gpurun_out/tl_r4f_rp2.txt | head -8
echo "== timeline cfg2 no P2"; PETAL_NO_P2=1 bash dev/tl.sh r4f_rp2_n dev/rpca_one.py; grep "^#" gpurun_out/tl_r4f_rp2_n.txt | head -5
echo "== timeline cfg4 P2"; bash dev/tl.sh r4f_rp4 dev/rpca_one.py cfg4; grep "^#" gpurun_out/tl_r4f_rp4.txt | head -8
echo "== timeline cfg4 no P2"; PETAL_NO_P2=1 bash dev/tl.sh r4f_rp4_n dev/rpca_one.py cfg4; grep "^#" gpurun_out/tl_r4f_rp4_n.txt | head -5
echo "== bench"; python bench.py --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/r4f_bench.json; python -c "
import json; r=json.load(open('gpurun_out/r4f_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['other_kernel'])"
echo "== parity"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5

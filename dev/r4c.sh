#!/bin/bash
# round 4, batch c: the ticketed re-basing kernel -- parity, timelines (fused vs not), bench
cd "$GRAFT_REPO_ROOT"
echo "== parity (rpca / gemm / spectrum / rank / determinism)"; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6
echo "== timeline cfg2 fused"; bash dev/tl.sh r4c_rp2 dev/rpca_one.py; tail -28 gpurun_out/tl_r4c_rp2.txt
echo "== timeline cfg2 unfused"; PETAL_NO_REBASE_FUSE=1 bash dev/tl.sh r4c_rp2_nf dev/rpca_one.py; grep "^#" gpurun_out/tl_r4c_rp2_nf.txt | head -3
echo "== timeline cfg4 fused"; bash dev/tl.sh r4c_rp4 dev/rpca_one.py cfg4; grep "^#" gpurun_out/tl_r4c_rp4.txt | head -14
echo "== timeline cfg4 unfused"; PETAL_NO_REBASE_FUSE=1 bash dev/tl.sh r4c_rp4_nf dev/rpca_one.py cfg4; grep "^#" gpurun_out/tl_r4c_rp4_nf.txt | head -3
echo "== bench"; python bench.py --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/r4c_bench.json; python -c "
import json; r=json.load(open('gpurun_out/r4c_bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'])"
echo "== bench K2 waves 2048"; PETAL_K2_WAVES=2048 python bench.py --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'])"
echo "== bench unfused"; PETAL_NO_REBASE_FUSE=1 python bench.py --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'])"

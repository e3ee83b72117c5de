#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_eigh.py -x -q -m gpu > gpurun_out/r6_h_eigh.txt 2>&1; tail -2 gpurun_out/r6_h_eigh.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r6_h_parity.txt 2>&1; tail -2 gpurun_out/r6_h_parity.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/r6_h_bench.json
python bench.py --gemm fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/r6_h_bench_fp32.json
python -c "
import json
for f in ('r6_h_bench','r6_h_bench_fp32'):
    d=json.load(open('gpurun_out/'+f+'.json')); print(f, d['ms_per_step'], d['serial_chain']['serial_chain_ms'])"
bash dev/tl.sh r6_h_rp2 dev/rpca_one.py > /dev/null 2>&1
grep -E "trieig|tridiag" gpurun_out/tl_r6_h_rp2.txt | head -3; head -1 gpurun_out/tl_r6_h_rp2.txt
./dev/eig_stress > gpurun_out/r6_h_eig_stress.txt 2>&1; tail -3 gpurun_out/r6_h_eig_stress.txt

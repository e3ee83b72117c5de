#!/bin/bash
# everything the round's README / profiles quote, in one GPU session (outputs under gpurun_out/final_*)
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/final_bench_default.log 2>&1; tail -1 gpurun_out/final_bench_default.log > gpurun_out/final_bench_default.json
bash dev/prof_fit.sh final > gpurun_out/final_prof_fit.txt 2>&1
python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/final_bench_cfg4.json
python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/final_bench_cfg5.json
python bench.py --gemm fp32 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/final_bench_fp32.json
python bench.py --gpus 2 --share-gpu --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/final_bench_cfg4_share2.json
python bench.py --gpus 2 --share-gpu --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/final_bench_cfg5_share2.json
bash dev/kt.sh final_ica "k_ica|k_atb_f64|tridiag|trieig|jacobi|eigh|k_sum|k_xp3|chol" dev/ica_bench.py > gpurun_out/final_ica_bench.txt 2>&1
bash dev/kt.sh final_ica5 "k_ica|k_atb_f64|tridiag|trieig|jacobi|eigh|k_sum|k_xp3|chol" dev/ica5_bench.py > gpurun_out/final_ica5_bench.txt 2>&1
bash dev/pmc_all.sh > gpurun_out/final_pmc_all.txt 2>&1
echo done

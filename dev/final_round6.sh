#!/bin/bash
# everything round 6's README / profiles quote, in one GPU session (outputs under gpurun_out/final6_*)
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/final6_bench_default.log 2> gpurun_out/final6_bench_default.err; tail -1 gpurun_out/final6_bench_default.log > gpurun_out/final6_bench_default.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final6_bench_driver.json
bash dev/prof_fit.sh final6 > gpurun_out/final6_prof_fit.txt 2>&1
cp gpurun_out/kt_final6/*/*_kernel_stats.csv gpurun_out/final6_kernel_stats.csv 2>/dev/null
python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final6_bench_cfg4.json
python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final6_bench_cfg5.json
python bench.py --gemm fp32 --steps 30 --warmup 5 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/final6_bench_fp32.json
python bench.py --config cfg4s --steps 5 --warmup 2 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/final6_bench_cfg4s_1gpu.json
python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final6_bench_cfg4s_share2.json
bash dev/kt.sh final6_ica "k_ica|k_atb_f64|k_gram3|k_presplit|tridiag|trieig|jacobi|eigh|k_sum|k_xp3|chol|symdecorr|ritz|whiten" dev/ica_bench.py > gpurun_out/final6_ica_bench.txt 2>&1
bash dev/kt.sh final6_ica5 "k_ica|k_atb_f64|k_gram3|k_presplit|tridiag|trieig|jacobi|eigh|k_sum|k_xp3|chol|symdecorr|ritz|whiten" dev/ica5_bench.py > gpurun_out/final6_ica5_bench.txt 2>&1
bash dev/tl.sh final6_rp2 dev/rpca_one.py > /dev/null 2>&1
bash dev/tl.sh final6_rp4 dev/rpca_one.py cfg4 > /dev/null 2>&1
bash dev/tl.sh final6_ica3 dev/ica_one.py > /dev/null 2>&1
bash dev/tl.sh final6_ica5 dev/ica_one.py cfg5 > /dev/null 2>&1
python dev/fit_ab.py > gpurun_out/final6_fit_ab_fused.log 2>&1
PETAL_NO_POW3=1 python dev/fit_ab.py > gpurun_out/final6_fit_ab_unfused.log 2>&1
# PMC passes (counters in their own runs, --kernel-trace only)
for mode in bf16x3; do for rows in 100000 1000000; do for counter in FETCH_SIZE WRITE_SIZE; do
  PETAL_GEMM=$mode dev/pmc_pass.sh tr_${mode}_${rows}_${counter} $rows $counter > /dev/null 2>&1
done; done; done
PMC_SCRIPT=dev/pmc_ica3.py dev/pmc_pass.sh ica3_j1 200000 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES > /dev/null 2>&1
python dev/pmc_table.py gpurun_out/pmc_ica3_j1 > gpurun_out/final6_pmc_ica_200000x32.txt 2>&1
dev/pmc_pass.sh pow3_j1 100000 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES > /dev/null 2>&1
python dev/pmc_table.py gpurun_out/pmc_pow3_j1 > gpurun_out/final6_pmc_pow3_100000x512.txt 2>&1
dev/pmc_pass.sh pow3_j2 100000 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY > /dev/null 2>&1
python dev/pmc_table.py gpurun_out/pmc_pow3_j2 >> gpurun_out/final6_pmc_pow3_100000x512.txt 2>&1
# keep only the counter CSVs of the traffic passes (small) for dev/pmc_traffic.py to read back in the build container
find gpurun_out/pmc_tr_* -type f ! -name "*counter_collection.csv" -delete 2>/dev/null
rm -rf gpurun_out/pmc_ica3_j1 gpurun_out/pmc_pow3_j1 gpurun_out/pmc_pow3_j2 gpurun_out/kt_*
# the phase stamps of the fused pass (debug library)
bash dev/build_dbg.sh > /dev/null 2>&1
python dev/pow3_phases.py > gpurun_out/final6_pow3_phases.txt 2>&1
# the sweeps of the round (dev/fuzz_round6.py: six kinds, both GEMM modes; the standing sweeps; eig_stress; the determinism soak)
timeout 1500 python dev/fuzz_round6.py 61 40 > gpurun_out/final6_fuzz_round6.txt 2>&1
FUZZ_GEMM=fp32 timeout 1200 python dev/fuzz_round6.py 62 25 > gpurun_out/final6_fuzz_round6_fp32.txt 2>&1
bash dev/fuzz_sweep.sh > gpurun_out/final6_fuzz_sweep.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o dev/eig_stress dev/eig_stress.hip petal-decomposition_amd/csrc/algo.cpp petal-decomposition_amd/csrc/api.cpp petal-decomposition_amd/csrc/rccl.cpp -ldl > /dev/null 2>&1
./dev/eig_stress > gpurun_out/final6_eig_stress.txt 2>&1
timeout 900 python dev/soak.py 1500 > gpurun_out/final6_soak.txt 2>&1
echo done

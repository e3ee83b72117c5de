#!/bin/bash
# everything round 3's README / profiles quote, in one GPU session (outputs under gpurun_out/final3_*)
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/final3_bench_default.log 2> gpurun_out/final3_bench_default.err; tail -1 gpurun_out/final3_bench_default.log > gpurun_out/final3_bench_default.json
bash dev/prof_fit.sh final3 > gpurun_out/final3_prof_fit.txt 2>&1
cp gpurun_out/kt_final3/*/*_kernel_stats.csv gpurun_out/final3_kernel_stats.csv 2>/dev/null
python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final3_bench_cfg4.json
python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final3_bench_cfg5.json
python bench.py --gemm fp32 --steps 30 --warmup 5 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 > gpurun_out/final3_bench_fp32.json
python bench.py --gpus 2 --share-gpu --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final3_bench_cfg4_share2.json
bash dev/kt.sh final3_ica "k_ica|k_atb_f64|tridiag|trieig|jacobi|eigh|k_sum|k_xp3|chol|symdecorr|ritz|whiten" dev/ica_bench.py > gpurun_out/final3_ica_bench.txt 2>&1
bash dev/kt.sh final3_ica5 "k_ica|k_atb_f64|tridiag|trieig|jacobi|eigh|k_sum|k_xp3|chol|symdecorr|ritz|whiten" dev/ica5_bench.py > gpurun_out/final3_ica5_bench.txt 2>&1
bash dev/tl.sh final3_rp2 dev/rpca_one.py > /dev/null 2>&1
bash dev/tl.sh final3_ica3 dev/ica_one.py > /dev/null 2>&1
# PMC passes (counters in their own runs, --kernel-trace only)
for mode in bf16x3; do for rows in 100000 1000000; do for counter in FETCH_SIZE WRITE_SIZE; do
  PETAL_GEMM=$mode dev/pmc_pass.sh tr_${mode}_${rows}_${counter} $rows $counter > /dev/null 2>&1
done; done; done
PMC_SCRIPT=dev/pmc_ica.py dev/pmc_pass.sh ica5_j1 500000 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES > /dev/null 2>&1
PMC_SCRIPT=dev/pmc_ica.py dev/pmc_pass.sh ica5_j2 500000 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES > /dev/null 2>&1
PMC_SCRIPT=dev/pmc_ica3.py dev/pmc_pass.sh ica3_j1 200000 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES > /dev/null 2>&1
PMC_SCRIPT=dev/pmc_ica3.py dev/pmc_pass.sh ica3_j2 200000 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES > /dev/null 2>&1
PMC_SCRIPT=dev/pmc_gram.py dev/pmc_pass.sh gram_j1 500000 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES > /dev/null 2>&1
python dev/pmc_table.py gpurun_out/pmc_ica5_j1 gpurun_out/pmc_ica5_j2 > gpurun_out/final3_pmc_ica_500000x64.txt 2>&1
python dev/pmc_table.py gpurun_out/pmc_ica3_j1 gpurun_out/pmc_ica3_j2 > gpurun_out/final3_pmc_ica_200000x32.txt 2>&1
python dev/pmc_table.py gpurun_out/pmc_gram_j1 > gpurun_out/final3_pmc_gram_500000x512.txt 2>&1
python dev/pmc_traffic.py r03 > gpurun_out/final3_pmc_traffic.txt 2>&1
cp profiles/r03_pmc_traffic.json gpurun_out/final3_pmc_traffic.json 2>/dev/null
rm -rf gpurun_out/pmc_* gpurun_out/kt_* 
echo done

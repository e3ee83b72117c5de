#!/bin/bash
# counters of the split-product Gram kernel (two --pmc passes, --kernel-trace only) at 500000 x 512 and 200000 x 256
cd "$GRAFT_REPO_ROOT"
for shape in "500000 512" "200000 256"; do
  set -- $shape
  tag=g5_$1x$2
  PMC_SCRIPT=dev/pmc_gram5.py dev/pmc_pass.sh ${tag}_j1 "$1 $2" GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES > /dev/null 2>&1
  PMC_SCRIPT=dev/pmc_gram5.py dev/pmc_pass.sh ${tag}_j2 "$1 $2" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY > /dev/null 2>&1
  PMC_SCRIPT=dev/pmc_gram5.py dev/pmc_pass.sh ${tag}_j3 "$1 $2" FETCH_SIZE > /dev/null 2>&1
  PMC_SCRIPT=dev/pmc_gram5.py dev/pmc_pass.sh ${tag}_j4 "$1 $2" WRITE_SIZE > /dev/null 2>&1
  for j in 1 2 3 4; do python dev/pmc_table.py gpurun_out/pmc_${tag}_j$j; done | grep "k_gram\|k_colsum\|k_colmean" > gpurun_out/pmc_gram5_$1x$2.txt 2>&1
  rm -rf gpurun_out/pmc_${tag}_j*
done

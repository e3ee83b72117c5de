#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the two power-iteration GEMM kernels, both modes, 100000 and 1e6 rows: eight separate --pmc passes
for mode in bf16x3 fp32; do
  for rows in 100000 1000000; do
    for counter in FETCH_SIZE WRITE_SIZE; do
      export PETAL_GEMM=$mode
      dev/pmc_pass.sh tr_${mode}_${rows}_${counter} $rows $counter
    done
  done
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for lib in petal-decomposition_amd/libpetal_hip.so dev/libpetal_nolowp.so; do
  echo "== $lib"
  export PETAL_HIP_LIBRARY=$PWD/$lib
  rm -rf gpurun_out/kt_t3 gpurun_out/kt_t5
  bash dev/kt.sh t3 "k_ica_tail|k_symdecorr" dev/ica_bench.py 2>&1 | grep "k_ica_tail\|k_symdecorr\|fit "
  bash dev/kt.sh t5 "k_ica_tail|k_symdecorr" dev/ica5_bench.py 2>&1 | grep "k_ica_tail\|k_symdecorr\|fit "
done
done
rm -rf gpurun_out/kt_t3 gpurun_out/kt_t5

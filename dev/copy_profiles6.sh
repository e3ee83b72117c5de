#!/bin/bash
# gpurun_out/final6_* (dev/final_round6.sh, one GPU session) -> profiles/r06_*
cd "$(dirname "$0")/.."
g=gpurun_out; p=profiles
cp $g/final6_bench_default.json $p/r06_bench_cfg2.json
cp $g/final6_bench_driver.json $p/r06_bench_cfg2_as_the_driver_runs_it.json
cp $g/final6_bench_fp32.json $p/r06_bench_cfg2_fp32_mode.json
cp $g/final6_bench_cfg4.json $p/r06_bench_cfg4_share.json
cp $g/final6_bench_cfg5.json $p/r06_bench_cfg5_share.json
cp $g/final6_bench_cfg4s_1gpu.json $p/r06_bench_cfg4s_strong_1gpu.json
cp $g/final6_bench_cfg4s_share2.json $p/r06_bench_cfg4s_strong_2ranks_sharing_one_gpu.json
cp $g/final6_kernel_stats.csv $p/r06_bench_cfg2_kernel_stats.csv
grep -v "amdgpu.ids\|^[EWI]2026" $g/final6_prof_fit.txt > $p/r06_bench_cfg2_kernel_summary.txt
grep -v "amdgpu.ids\|^[EWI]2026" $g/final6_ica_bench.txt > $p/r06_fastica_kernel_stats.txt
echo "---- one rank's share of configs[4] (500000 x 512, 64 components)" >> $p/r06_fastica_kernel_stats.txt
grep -v "amdgpu.ids\|^[EWI]2026" $g/final6_ica5_bench.txt >> $p/r06_fastica_kernel_stats.txt
(echo "# fused passes on (default)"; grep -v amdgpu.ids $g/final6_fit_ab_fused.log; echo "# PETAL_NO_POW3=1: K1 + K2"; grep -v amdgpu.ids $g/final6_fit_ab_unfused.log) > $p/r06_fit_fused_vs_unfused.txt
grep -v amdgpu.ids $g/final6_pow3_phases.txt > $p/r06_k_pow3_phase_stamps.txt
cp $g/final6_pmc_ica_200000x32.txt $p/r06_pmc_fastica_200000x32.txt
cp $g/final6_pmc_pow3_100000x512.txt $p/r06_pmc_k_pow3_100000x512.txt
cp $g/tl_final6_rp2.txt $p/r06_timeline_rpca_cfg2.txt
cp $g/tl_final6_rp4.txt $p/r06_timeline_rpca_cfg4_share.txt
cp $g/tl_final6_ica3.txt $p/r06_timeline_fastica_cfg3.txt
cp $g/tl_final6_ica5.txt $p/r06_timeline_fastica_cfg5_share.txt
cp $g/final6_eig_stress.txt $p/r06_eig_stress.txt
grep -v amdgpu.ids $g/final6_soak.txt > $p/r06_soak_1500.txt
(echo "# dev/fuzz_sweep.sh (the standing sweeps: fuzz_all 11 60 ok-count + FAIL lines, fuzz_all fp32-mode 12 30, fuzz_rpca 13 60, fuzz_rank, fuzz_clip)"; grep -v amdgpu.ids $g/final6_fuzz_sweep.txt | cut -c1-240) > $p/r06_fuzz_sweep_summary.txt
(echo "# dev/fuzz_round6.py 61 40 (default mode): $(grep -c '^ok' $g/final6_fuzz_round6.txt) ok; every miss with the ORACLE IN FLOAT32 on the same input beside it"; grep "FAIL\|failures" $g/final6_fuzz_round6.txt | cut -c1-260
 echo "# FUZZ_GEMM=fp32 dev/fuzz_round6.py 62 25: $(grep -c '^ok' $g/final6_fuzz_round6_fp32.txt) ok"; grep "FAIL\|failures" $g/final6_fuzz_round6_fp32.txt | cut -c1-260) > $p/r06_fuzz_round6.txt
python dev/pmc_traffic.py r06 > /dev/null 2>&1
ls -la $p/r06_* | wc -l

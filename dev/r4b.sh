#!/bin/bash
# round 4, batch b: Gram min-waves A/B, host_in pitch A/B, new bench records, sharded tests
cd "$GRAFT_REPO_ROOT"
echo "== gram minwaves 4 (default)"; bash dev/kt.sh r4b_g4 "k_atb_f64" dev/gram_bench.py 500000
echo "== gram minwaves 1 (old)"; PETAL_HIP_LIBRARY=dev/libpetal_gram1.so bash dev/kt.sh r4b_g1 "k_atb_f64" dev/gram_bench.py 500000
echo "== gram 200000x256 (Pca d=512 only; use ica_bench for d=256)"; bash dev/kt.sh r4b_i4 "k_atb_f64" dev/ica_bench.py
PETAL_HIP_LIBRARY=dev/libpetal_gram1.so bash dev/kt.sh r4b_i1 "k_atb_f64" dev/ica_bench.py
echo "== bench default"; python bench.py --no-cpu-baseline 2> gpurun_out/r4b_bench.err | tail -1 > gpurun_out/r4b_bench.json; tail -3 gpurun_out/r4b_bench.err
echo "== host_in without row pad"; PETAL_NO_ROW_PAD=1 python bench.py --no-cpu-baseline --no-northstar --steps 20 2>/dev/null | tail -1 > gpurun_out/r4b_bench_nopad.json
echo "== bench --gpus 2 --share-gpu (cfg4s strong)"; python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r4b_share2.err | tail -1 > gpurun_out/r4b_share2.json; tail -3 gpurun_out/r4b_share2.err
echo "== bench --gpus 1 --config cfg4s"; python bench.py --config cfg4s --steps 5 --warmup 2 --no-cpu-baseline --no-northstar 2> gpurun_out/r4b_cfg4s.err | tail -1 > gpurun_out/r4b_cfg4s.json; tail -3 gpurun_out/r4b_cfg4s.err
echo "== sharded + contract tests"; python -m pytest tests/test_gpu_sharded.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -5
rm -rf gpurun_out/kt_r4b_*/

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/final6_bench_default.log 2> gpurun_out/final6_bench_default.err; tail -1 gpurun_out/final6_bench_default.log > gpurun_out/final6_bench_default.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final6_bench_driver.json
python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final6_bench_cfg4.json
python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final6_bench_cfg5.json
python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final6_bench_cfg4s_share2.json
for f in default driver cfg4 cfg5 cfg4s_share2; do python -c "
import json
d=json.load(open('gpurun_out/final6_bench_$f.json')); print('$f', d.get('ms_per_step'), (d.get('serial_chain') or {}).get('serial_chain_ms'), (d.get('roofline') or {}).get('frac'))"; done

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONFAULTHANDLER=1
for rep in 1 2 3 4 5 6; do
  setsid python bench.py --steps 3 --warmup 1 --no-northstar --gpus 2 --share-gpu --no-cpu-baseline > gpurun_out/r6_x_out$rep.txt 2> gpurun_out/r6_x_err$rep.txt &
  pid=$!
  for i in $(seq 1 30); do sleep 3; if ! kill -0 $pid 2>/dev/null; then break; fi; done
  if kill -0 $pid 2>/dev/null; then
    echo "rep $rep: still running after 90 s: dumping stacks"
    pgid=$(ps -o pgid= -p $pid | tr -d ' ')
    kill -ABRT -- -$pgid
    sleep 3
    grep -v "amdgpu.ids" gpurun_out/r6_x_err$rep.txt | tail -80 | cut -c1-200
    break
  else
    echo "rep $rep: finished; $(tail -c 100 gpurun_out/r6_x_out$rep.txt)"
  fi
done
timeout 1200 python -m pytest tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -3

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONFAULTHANDLER=1
setsid python bench.py --steps 3 --warmup 1 --no-northstar --gpus 2 --share-gpu --no-cpu-baseline > gpurun_out/r6_w_out.txt 2> gpurun_out/r6_w_err.txt &
pid=$!
for i in $(seq 1 24); do sleep 5; if ! kill -0 $pid 2>/dev/null; then break; fi; done
if kill -0 $pid 2>/dev/null; then
  echo "still running after 120 s: dumping stacks"
  pgid=$(ps -o pgid= -p $pid | tr -d ' ')
  kill -ABRT -- -$pgid
  sleep 3
fi
tail -c 600 gpurun_out/r6_w_out.txt
grep -v "amdgpu.ids" gpurun_out/r6_w_err.txt | tail -60 | cut -c1-200

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PETAL_DEBUG_ATTEMPTS=1
for rep in 1 2 3 4 5 6 7 8; do
  python bench.py --steps 3 --warmup 1 --no-northstar --no-cpu-baseline > /dev/null 2>&1
  setsid python bench.py --steps 3 --warmup 1 --no-northstar --gpus 2 --share-gpu --no-cpu-baseline > gpurun_out/r6_aa_out$rep.txt 2> gpurun_out/r6_aa_err$rep.txt &
  pid=$!
  for i in $(seq 1 30); do sleep 2; if ! kill -0 $pid 2>/dev/null; then break; fi; done
  if kill -0 $pid 2>/dev/null; then
    echo "rep $rep: still running after 60 s"
    pgid=$(ps -o pgid= -p $pid | tr -d ' ')
    kill -KILL -- -$pgid
    sleep 2
    grep "petal rank-offset" gpurun_out/r6_aa_err$rep.txt | tail -40 | cut -c1-220
    break
  else
    echo "rep $rep: finished; $(grep -c 'petal rank-offset' gpurun_out/r6_aa_err$rep.txt) attempt lines; $(grep 'petal rank-offset' gpurun_out/r6_aa_err$rep.txt | grep -v 'code 0.0' | head -3)"
  fi
done

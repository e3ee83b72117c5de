#!/bin/bash
# A/B of the dot2 form of split3 (dev/libpetal_dot2.so) against the built library, alternating on one box
cd "$GRAFT_REPO_ROOT"
./dev/micro_dot2
for rep in 1 2; do
  for lib in petal-decomposition_amd/libpetal_hip.so dev/libpetal_dot2.so; do
    echo "== $lib"
    PETAL_HIP_LIBRARY=$PWD/$lib python dev/fit_ab.py 2>&1 | grep -v amdgpu.ids
    PETAL_HIP_LIBRARY=$PWD/$lib python dev/pow3_bench.py 2>&1 | grep -v amdgpu.ids
  done
done
for lib in petal-decomposition_amd/libpetal_hip.so dev/libpetal_dot2.so; do
  echo "== $lib"
  PETAL_HIP_LIBRARY=$PWD/$lib python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('cfg4', r['ms_per_step'])"
  PETAL_HIP_LIBRARY=$PWD/$lib python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('cfg5', r['ms_per_step'])"
  PETAL_HIP_LIBRARY=$PWD/$lib python dev/ica_bench.py 2>&1 | grep -v amdgpu.ids | tail -3
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for lib in petal-decomposition_amd/libpetal_hip.so dev/libpetal_sp4.so; do
  PETAL_HIP_LIBRARY=$PWD/$lib python dev/fit_ab1.py 2>&1 | grep -v amdgpu
done; done
python dev/steer_err.py 2>&1 | grep -v amdgpu
python -m pytest tests -q -m gpu -x -k "steering or rpca_parity or low_iter or two_plane" 2>&1 | tail -3
for lib in petal-decomposition_amd/libpetal_hip.so dev/libpetal_sp4.so; do
  PETAL_HIP_LIBRARY=$PWD/$lib python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-northstar 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('cfg4', r['ms_per_step'], r.get('rpca_redo'))"
done

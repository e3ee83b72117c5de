#!/bin/bash
# The product's host algorithms (api.cpp, algo.cpp, rccl.cpp) on the host simulation of the device ops, under AddressSanitizer +
# UndefinedBehaviorSanitizer (CPU only: the GPU pool has no sanitizer support): tests/test_hostsim.py against that build.
cd "$(dirname "$0")/.."
mkdir -p tests/_build
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o tests/_build/libpetal_hostsim_asan.so \
  petal-decomposition_amd/csrc/api.cpp petal-decomposition_amd/csrc/algo.cpp petal-decomposition_amd/csrc/rccl.cpp oracle/cpu_ops.cpp -ldl || exit 1
export PETAL_HOSTSIM_LIBRARY=$PWD/tests/_build/libpetal_hostsim_asan.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
python -m pytest tests/test_hostsim.py -x -q -p no:cacheprovider "$@" > tests/_build/asan.log 2>&1; grep -c "runtime error\|AddressSanitizer" tests/_build/asan.log; grep "runtime error\|AddressSanitizer" tests/_build/asan.log | sort | uniq -c | head -10; tail -2 tests/_build/asan.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== alternating fits"; timeout 900 python dev/alt_fits.py 2>&1 | tail -5
echo "== new test"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "alternating or two_plane" 2>&1 | grep -E "passed|failed|Error" | tail -3
echo "== rpca crowded-spectrum cases with and without the two-plane iterate (fuzz_all seed 22, rpca only)"
timeout 900 python dev/fuzz_all.py 22 50 rpca 2>&1 | grep -c "^ok"; timeout 900 python dev/fuzz_all.py 22 50 rpca 2>&1 | grep "^FAIL"
PETAL_NO_P2=1 timeout 900 python dev/fuzz_all.py 22 50 rpca 2>&1 | grep "^FAIL"
echo "== fuzz_clip"; timeout 600 python dev/fuzz_clip.py 2>&1 | grep -v "^ok" | tail -5

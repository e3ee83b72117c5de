#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for seed in 21 22; do
  echo "== fuzz_all seed $seed (split-product)"; timeout 1200 python dev/fuzz_all.py $seed 50 2>&1 | grep -v "^ok" | tail -15
done
echo "== fuzz_all seed 23 (fp32 mfma)"; FUZZ_GEMM=fp32 timeout 900 python dev/fuzz_all.py 23 30 2>&1 | grep -v "^ok" | tail -10
echo "== fuzz_rank"; timeout 600 python dev/fuzz_rank.py 2>&1 | grep -v "^ok" | tail -10
echo "== fuzz_clip"; timeout 600 python dev/fuzz_clip.py 2>&1 | grep -v "^ok" | tail -10
echo "== alternating fits on one ctx (fork / join, pool reuse)"; timeout 600 python dev/alt_fits.py 2>&1 | tail -5

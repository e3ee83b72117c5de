#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash dev/kt.sh t3 "k_ica_tail|k_ica3p|k_ica_reduce|k_symdecorr" dev/ica_bench.py 2>&1 | grep -v "^E2026\|^W2026\|amdgpu.ids"
bash dev/kt.sh t5 "k_ica_tail|k_ica3p|k_ica_reduce|k_symdecorr|k_gram" dev/ica5_bench.py 2>&1 | grep -v "^E2026\|^W2026\|amdgpu.ids"
python -m pytest tests -q -m gpu -k "ica or Ica" 2>&1 | tail -3

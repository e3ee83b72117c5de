#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== timeline cfg2 fused (batched loads)"; bash dev/tl.sh r4e_rp2 dev/rpca_one.py; grep "^#" gpurun_out/tl_r4e_rp2.txt | head -8
echo "== timeline cfg4 fused"; bash dev/tl.sh r4e_rp4 dev/rpca_one.py cfg4; grep "^#" gpurun_out/tl_r4e_rp4.txt | head -8
echo "== fastica ticket"; bash dev/tl.sh r4e_ica dev/ica_one.py; grep "^#" gpurun_out/tl_r4e_ica.txt | head -12
echo "== fastica no ticket"; PETAL_NO_ICA_TICKET=1 bash dev/tl.sh r4e_ica_nt dev/ica_one.py; grep "^#" gpurun_out/tl_r4e_ica_nt.txt | head -12
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ica" 2>&1 | grep -E "passed|failed"

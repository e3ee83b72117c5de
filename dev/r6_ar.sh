#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for s in 101 102; do FUZZ6_ONLY=wide timeout 900 python dev/fuzz_round6.py $s 40 2>&1 | grep "FAIL\|failures" | cut -c1-230; done
FUZZ6_ONLY=wide FUZZ_GEMM=fp32 timeout 900 python dev/fuzz_round6.py 103 40 2>&1 | grep "FAIL\|failures" | cut -c1-230
python - <<'PY'
import time, numpy as np, torch, sys
sys.path.insert(0, "tests")
import petal_decomposition_amd as petal
from oracle import petal_oracle as po
ctx = petal.Context(0)
x = torch.from_numpy(po.synth_pca(100000, 512, 64, seed=2, dtype=np.float32)).cuda()
om = np.random.default_rng(3).standard_normal((512, 74)).astype(np.float32)
for it in (3, 4, 5):
    m = petal.RandomizedPca(64, ctx=ctx, n_iter=it)
    for _ in range(5): m.fit(x, omega=om)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.fit(x, omega=om)
    torch.cuda.synchronize(); print(f"configs[1] shape, n_iter={it}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per fit")
PY

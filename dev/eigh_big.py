"""dev: time and check the two-stage eigen-solver at orders beyond the LDS-resident kernels (Pca with k = d)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import petal_decomposition_amd as petal
ctx = petal.Context(0)
for d in [int(a) for a in sys.argv[1:]] or [200, 300, 514, 600]:
    rng = np.random.default_rng(d)
    n = 4 * d
    u, _ = np.linalg.qr(rng.standard_normal((n, d)))
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    s = 1.0 - 0.9 * np.arange(d) / d
    x = np.ascontiguousarray(((u * s) @ q.T * np.sqrt(n)))
    m = petal.Pca(d, ctx=ctx)
    t0 = time.perf_counter(); m.fit(x); t1 = time.perf_counter()
    xc = x - x.mean(0)
    ref = np.linalg.svd(xc, compute_uv=False)
    sg = np.asarray(m.singular_values()); c = np.asarray(m.components())
    live = sg > 1e-10 * sg[0]
    print(d, "fit %.3f s" % (t1 - t0), "sigma err %.2e" % np.abs(sg[live] ** 2 - ref[:live.sum()] ** 2).max(),
          "orth %.2e" % np.abs(c[live] @ c[live].T - np.eye(live.sum())).max(), flush=True)

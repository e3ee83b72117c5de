import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, petal_decomposition_amd as petal
from oracle import petal_oracle as po
ctx = petal.Context(0)
for dt in (np.float32, np.float64):
    for what in ("nan", "inf", "zeros", "const"):
        x = po.synth_pca(20000, 512, 5, seed=1, dtype=dt)
        if what == "nan": x[17, 3] = np.nan
        if what == "inf": x[17, 3] = np.inf
        if what == "zeros": x[:] = 0
        if what == "const": x[:] = 3.5
        for name, mk in (("rpca", lambda: petal.RandomizedPca(5, ctx=ctx, n_iter=4)), ("pca", lambda: petal.Pca(5, ctx=ctx)), ("ica", lambda: petal.FastIca(ctx=ctx, n_components=5))):
            t0 = time.time()
            try:
                m = mk(); m.fit(x)
                comp = m.components() if callable(getattr(m, "components")) else m.components
                r = f"returned; finite components: {bool(np.isfinite(np.asarray(comp)).all())}"
            except Exception as e:
                r = f"raised {type(e).__name__}: {str(e)[:80]}"
            print(f"{dt.__name__} {what} {name}: {r} ({time.time() - t0:.2f} s)", flush=True)

"""cfg 5 prediction (V <- V L^-T over all of L, M = 10 000 rows): ms vs the panel width of the substitution."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n, d, m = 65536, 16, 10000
X, y = syn.make_dataset(n, d)
th = syn.default_theta("matern52_ard", d, dtype="f32")
Xs = syn.make_test_points(m, d)
h = _lib.Handle(X, y, "matern52_ard", dtype=32)
assert h.fit(th) == 0
for panel in (4, 6, 8, 12, 16, 4):
    h.set_option("panel", panel)
    h.predict(Xs[:256])
    t0 = time.perf_counter(); mu, var = h.predict(Xs); dt = time.perf_counter() - t0
    print(f"predict panel={panel}: {dt*1e3:.1f} ms ({n*n*m/dt/1e12:.1f} TFLOP/s) mu0={mu[0]:.6f}", flush=True)
h.close()

import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
X, y = syn.make_dataset(n, 8)
th = syn.default_theta("se_ard", 8)
h = _lib.Handle(X, y, "se_ard")
h.loglik_grad(th)
t0 = time.perf_counter(); ll, g, info = h.loglik_grad(th); print("grad ms", (time.perf_counter() - t0) * 1e3)
h.close()

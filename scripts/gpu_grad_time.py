import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n in (4096, 8192, 16384):
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    h.loglik_grad(th)
    t0 = time.perf_counter(); ll, g, info = h.loglik_grad(th); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); h.loglik(th); dl = time.perf_counter() - t0
    print(f"N={n}: loglik {dl*1e3:.1f} ms, loglik+grad {dt*1e3:.1f} ms (x{dt/dl:.1f}); |grad|max={abs(g).max():.3g}", flush=True)
    h.close()

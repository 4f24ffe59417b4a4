"""A/B of the gemm wave grid (2x2 vs 4x4) for the panel-stream launches and the trailing SYRK."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n, d, kernel, reps in ((512, 1, "se", 20), (4096, 8, "se_ard", 10), (32768, 8, "se_ard", 3)):
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    h = _lib.Handle(X, y, kernel)
    for pw, sw in ((4, 4), (16, 4), (0, 4), (16, 16), (4, 16)):
        h.set_option("panel_waves", pw); h.set_option("syrk_waves", sw)
        h.loglik(th)
        t0 = time.perf_counter()
        for _ in range(reps):
            ll, info = h.loglik(th)
        dt = (time.perf_counter() - t0) / reps
        print(f"N={n} panel_waves={pw} syrk_waves={sw}: {dt*1e3:.3f} ms/eval ll={ll:.10g}", flush=True)
    h.close()

"""cfg 5 fit (fp32 Matern N=65536 d=16): ms vs base panel width with the wide-early-panel rule on."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n, d = 65536, 16
X, y = syn.make_dataset(n, d)
th = syn.default_theta("matern52_ard", d, dtype="f32")
h = _lib.Handle(X, y, "matern52_ard", dtype=32)
h.loglik(th)
for panel, wide in ((4, 1), (4, 0), (6, 1), (8, 1), (8, 0), (4, 1)):
    h.set_option("panel", panel); h.set_option("panel_wide", wide)
    h.loglik(th)
    t0 = time.perf_counter()
    for _ in range(2):
        ll, info = h.loglik(th)
    dt = (time.perf_counter() - t0) / 2
    print(f"fp32 N={n}: panel={panel} wide={wide}: {dt*1e3:.1f} ms ({n**3/3/dt/1e12:.1f} TFLOP/s) ll={ll:.6g}", flush=True)
h.close()

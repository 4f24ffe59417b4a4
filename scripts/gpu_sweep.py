"""Option sweep at N=32768 (panel width, swizzle, look-ahead)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y = syn.make_dataset(n, 8)
th = syn.default_theta("se_ard", 8)
h = _lib.Handle(X, y, "se_ard")
h.loglik(th)
for panel, swz, la in ((4, 1, 1), (8, 1, 1), (6, 1, 1), (2, 1, 1), (4, 0, 1), (8, 0, 1), (4, 1, 0), (8, 1, 0)):
    h.set_option("panel", panel); h.set_option("xcd_swizzle", swz); h.set_option("lookahead", la)
    h.loglik(th)
    t0 = time.perf_counter()
    for _ in range(3):
        ll, info = h.loglik(th)
    dt = (time.perf_counter() - t0) / 3
    print(f"N={n} panel={panel} swizzle={swz} lookahead={la}: {dt*1e3:.2f} ms/eval  ({n**3/3/dt/1e12:.2f} TFLOP/s) ll={ll:.9g}", flush=True)

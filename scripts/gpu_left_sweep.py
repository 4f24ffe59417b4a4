"""panel_left (-1 auto / 1 left-looking in-panel updates) vs N, one theta."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n in (14000, 16384, 24576, 32768, 49152):
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    h.loglik(th)
    row = []
    for left in (-1, 1, -1, 1):
        h.set_option("panel_left", left)
        h.loglik(th)
        reps = 4 if n <= 24576 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ll, info = h.loglik(th)
        row.append(f"left={left}: {(time.perf_counter()-t0)/reps*1e3:7.2f}")
    print(f"N={n}: " + "  ".join(row), flush=True)
    h.close()

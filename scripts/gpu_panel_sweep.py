"""Outer panel width (in 128-tiles) vs N for one theta at a time: ms per evaluation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n in (12288, 16384, 24576, 32768, 49152):
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    h.loglik(th)
    row = []
    for panel in (3, 4, 5, 6, 8, 4):
        h.set_option("panel", panel)
        h.loglik(th)
        reps = 4 if n <= 24576 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ll, info = h.loglik(th)
        dt = (time.perf_counter() - t0) / reps
        row.append(f"P={panel}: {dt*1e3:7.2f}")
    print(f"N={n}: " + "  ".join(row), flush=True)
    h.close()

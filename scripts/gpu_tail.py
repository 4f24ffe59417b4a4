"""Tail cut-over sweep: ms/eval vs dataflow_tail (tile columns handed to the dataflow kernel)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n in (16384, 24576, 32768):
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    h.loglik(th)
    for tail in (0, 32, 64, 80, 96, 64):
        h.set_option("dataflow_tail", tail)
        h.loglik(th)
        reps = 6 if n <= 16384 else 4
        t0 = time.perf_counter()
        for _ in range(reps):
            ll, info = h.loglik(th)
        dt = (time.perf_counter() - t0) / reps
        print(f"N={n} tail={tail}: {dt*1e3:.2f} ms  ll={ll:.10g}", flush=True)
    h.close()

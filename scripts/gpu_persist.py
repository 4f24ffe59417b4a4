"""A/B of the persistent trailing SYRK (work counter, reserved residency slots) across N."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n, reps in ((4096, 10), (8192, 6), (16384, 4), (32768, 3)):
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    for ps, rs in ((0, 0), (1, 0), (1, 16), (1, 24), (1, 48), (1, 96)):
        h.set_option("persistent_syrk", ps); h.set_option("reserve_slots", rs)
        h.loglik(th)
        t0 = time.perf_counter()
        for _ in range(reps):
            ll, info = h.loglik(th)
        dt = (time.perf_counter() - t0) / reps
        print(f"N={n} persistent={ps} reserve_slots={rs}: {dt*1e3:.3f} ms/eval ll={ll:.12g}", flush=True)
    h.close()

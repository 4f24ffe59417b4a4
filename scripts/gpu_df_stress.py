"""Stress of the dataflow schedule: thousands of back-to-back evaluations must reproduce bit-identical
values per theta (a stale-cache or ordering bug would show as an occasional mismatch or an abort)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bayesianinference_amd import _lib, synthetic as syn
for n, d, fine, reps, B in ((640, 3, 16, 3000, 1), (640, 3, 0, 2000, 1), (1500, 5, 16, 1500, 1), (4096, 8, 0, 300, 1), (900, 2, 16, 500, 8), (2048, 4, 0, 200, 4)):
    X, y = syn.make_dataset(n, d)
    h = _lib.Handle(X, y, "se_ard")
    h.set_option("dataflow_fine_nt", fine)
    base = syn.default_theta("se_ard", d)
    ths = [base * (1 + 0.1 * k) for k in range(3)]
    if B == 1:
        ref = [h.loglik_parts(t) for t in ths]
    else:
        Ths = [np.stack([t * (1 + 0.01 * s) for s in range(B)]) for t in ths]
        ref = [h.loglik_batch(T)[0].copy() for T in Ths]
    bad = 0
    t0 = time.perf_counter()
    for r in range(reps):
        k = r % 3
        if B == 1:
            bad += h.loglik_parts(ths[k]) != ref[k]
        else:
            bad += not np.array_equal(h.loglik_batch(Ths[k])[0], ref[k])
    dt = time.perf_counter() - t0
    print(f"N={n} fine_nt={fine} B={B}: {reps} evaluations, {bad} mismatches, {dt/reps*1e3:.3f} ms each", flush=True)
    h.close()

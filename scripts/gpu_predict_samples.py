"""Batched mixture prediction (BGP:343-376): S posterior samples x M test points at small N."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bayesianinference_amd import _lib, synthetic as syn
for n, d, S, M in ((512, 1, 500, 100), (512, 1, 2000, 200), (2048, 4, 300, 500), (4096, 8, 200, 1000)):
    kernel = "se" if d == 1 else "se_ard"
    X, y = syn.make_dataset(n, d)
    Th = syn.theta_batch(S, kernel, d); Th[:, -1] = np.maximum(Th[:, -1], 0.05)
    Xs = syn.make_test_points(M, d)
    h = _lib.Handle(X, y, kernel)
    h.predict_samples(Th, Xs)                             # warm-up sizes the S-slot workspace
    t0 = time.perf_counter(); mean, var, info = h.predict_samples(Th, Xs); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); h.loglik_batch(Th); dl = time.perf_counter() - t0
    print(f"N={n} S={S} M={M}: predict_samples {dt*1e3:.1f} ms ({S/dt:.0f} samples/s); factor-only batch {dl*1e3:.1f} ms; bad={int((info!=0).sum())}", flush=True)
    h.close()

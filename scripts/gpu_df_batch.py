"""Dataflow vs multi-kernel schedule for theta batches (cfg 4 shape) and fp32 (developer check)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bayesianinference_amd import _lib, synthetic as syn
for n, d, B, dtype in ((256, 2, 16, 64), (256, 2, 32, 64), (256, 2, 64, 64), (512, 1, 16, 64), (512, 1, 32, 64), (512, 1, 64, 64), (1024, 3, 16, 64), (1024, 3, 32, 64), (2048, 8, 16, 64)):
    kernel = "se" if d == 1 else "se_ard"
    X, y = syn.make_dataset(n, d)
    Th = syn.theta_batch(B, kernel, d)
    h = _lib.Handle(X, y, kernel, dtype=dtype)
    out = {}
    for df in (0, 1):
        h.set_option("dataflow", df)
        h.set_option("dataflow_max_slots", 100000)
        h.loglik_batch(Th)
        t0 = time.perf_counter()
        ll, info = h.loglik_batch(Th)
        out[df] = (ll, info, time.perf_counter() - t0)
    ok = np.array_equal(out[0][1], out[1][1])
    good = out[0][1] == 0
    diff = float(np.max(np.abs(out[0][0][good] - out[1][0][good]) / np.abs(out[0][0][good]))) if good.any() else 0.0
    print(f"N={n} B={B} f{dtype}: multi-kernel {B/out[0][2]:.0f} evals/s  dataflow {B/out[1][2]:.0f} evals/s | info equal {ok} fails {int((~good).sum())} max rel diff {diff:.2e}", flush=True)
    h.close()

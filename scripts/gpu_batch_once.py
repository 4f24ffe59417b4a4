"""cfg 4 once, for profiling: 200 thetas x N=4096 through gphip_loglik_batch (one warm-up call, two measured)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
X, y = syn.make_dataset(4096, 8)
Th = syn.theta_batch(200, "se_ard", 8)
Th[:, -1] = np.maximum(Th[:, -1], 0.05)
h = _lib.Handle(X, y, "se_ard")
h.loglik_batch(Th)
t0 = time.perf_counter()
for _ in range(2):
    out, info = h.loglik_batch(Th)
dt = (time.perf_counter() - t0) / 2
print(f"200 x N=4096: {dt*1e3:.1f} ms, {200/dt:.0f} evals/s, {200*4096**3/3/dt/1e12:.1f} TFLOP/s, failed={int((info != 0).sum())}")
h.close()

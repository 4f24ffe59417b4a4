"""Per-class kernel time of a BATCH evaluation (cfg 4: 200 theta x N=4096) next to its wall time.
   python scripts/gpu_classes_batch.py [B [N [opt=val ...]]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
X, y = syn.make_dataset(n, 8)
Th = syn.theta_batch(B, "se_ard", 8)
Th[:, -1] = np.maximum(Th[:, -1], 0.05)
h = _lib.Handle(X, y, "se_ard")
for a in sys.argv[3:]:
    k, v = a.split("=")
    h.set_option(k, int(v))
h.loglik_batch(Th[:8]); h.loglik_batch(Th)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); h.loglik_batch(Th); best = min(best, time.perf_counter() - t0)
h.set_option("profile", 2); h.loglik_batch(Th); h.reset_profile()
t0 = time.perf_counter(); h.loglik_batch(Th); wallp = (time.perf_counter() - t0) * 1e3
pr = h.profile()
print(f"B={B} N={n} {' '.join(sys.argv[3:])}: wall {best*1e3:.2f} ms = {B * n**3 / 3 / best / 1e12:.1f} TFLOP/s (profiled run {wallp:.2f}) | " +
      " ".join(f"{k}={v['ms']:.2f}ms/{int(v['launches'])}" + (f"({v['flops']/v['ms']/1e9:.1f}TF)" if v['flops'] > 0 and v['ms'] > 0 else "")
               for k, v in pr.items() if v["launches"]), flush=True)

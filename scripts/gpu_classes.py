"""Per-class kernel time (HIP events around every launch, option profile = 2) next to the wall time of an evaluation.
   python scripts/gpu_classes.py N [opt=val ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = int(sys.argv[1])
X, y = syn.make_dataset(n, 8)
th = syn.default_theta("se_ard", 8)
h = _lib.Handle(X, y, "se_ard")
for a in sys.argv[2:]:
    k, v = a.split("=")
    h.set_option(k, int(v))
h.loglik(th); h.loglik(th)
t0 = time.perf_counter()
for _ in range(5):
    h.loglik(th)
wall = (time.perf_counter() - t0) / 5 * 1e3
h.set_option("profile", 2); h.loglik(th); h.reset_profile()
t0 = time.perf_counter(); h.loglik(th); wallp = (time.perf_counter() - t0) * 1e3
pr = h.profile()
print(f"N={n} {' '.join(sys.argv[2:])}: wall {wall:.2f} ms (profiled run {wallp:.2f}) | " +
      " ".join(f"{k}={v['ms']:.2f}ms/{int(v['launches'])}" + (f"({v['flops']/v['ms']/1e9:.1f}TF)" if v['flops'] > 0 and v['ms'] > 0 else "")
               for k, v in pr.items() if v["launches"]), flush=True)

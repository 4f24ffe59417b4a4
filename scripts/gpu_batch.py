"""cfg-4 style probe: B theta at N=4096 in one batched call vs one by one."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 200
X, y = syn.make_dataset(n, 8)
Th = syn.theta_batch(B, "se_ard", 8)
Th[:, -1] = np.maximum(Th[:, -1], 0.05)
h = _lib.Handle(X, y, "se_ard")
for panel in (4, 2, 8):
    h.set_option("panel", panel)
    h.loglik_batch(Th[:8])
    out, info = h.loglik_batch(Th)
    t0 = time.perf_counter(); out, info = h.loglik_batch(Th); dt = time.perf_counter() - t0
    print(f"N={n} B={B} panel={panel}: batched {dt*1e3:.1f} ms total, {B/dt:.1f} evals/s, {B*n**3/3/dt/1e12:.2f} TFLOP/s, bad={int((info!=0).sum())}", flush=True)
h.set_option("panel", 4)
t0 = time.perf_counter()
for i in range(20):
    h.loglik(Th[i])
dt = (time.perf_counter() - t0) / 20
print(f"N={n} single: {dt*1e3:.2f} ms/eval, {1/dt:.1f} evals/s, {n**3/3/dt/1e12:.2f} TFLOP/s")
h.set_option("profile", 2); h.reset_profile(); h.loglik_batch(Th)
for k, v in h.profile().items():
    if v["launches"]:
        print(f"   {k:14s} {v['ms']:9.3f} ms  {int(v['launches']):5d} launches  {v['flops']/max(v['ms'],1e-9)/1e9:8.2f} TFLOP/s")

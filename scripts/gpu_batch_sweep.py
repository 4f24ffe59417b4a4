"""cfg-4 option sweep: B thetas x N=4096 through gphip_loglik_batch for look-ahead on/off, panel widths and
left/right-looking in-panel updates; per-class profile of the best and of the default."""
import os, sys, time, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X, y = syn.make_dataset(n, 8)
h = _lib.Handle(X, y, "se_ard")
for B in (200, 32):
    Th = syn.theta_batch(B, "se_ard", 8)
    Th[:, -1] = np.maximum(Th[:, -1], 0.05)
    h.loglik_batch(Th)
    for la, panel, left in itertools.product((1, 0), (2, 4, 8), (-1, 0)):
        h.set_option("lookahead", la); h.set_option("panel", panel); h.set_option("panel_left", left)
        h.loglik_batch(Th)
        t0 = time.perf_counter()
        for _ in range(3):
            out, info = h.loglik_batch(Th)
        dt = (time.perf_counter() - t0) / 3
        print(f"N={n} B={B} lookahead={la} panel={panel} left={left}: {dt*1e3:7.2f} ms  {B/dt:7.1f} evals/s  {B*n**3/3/dt/1e12:6.2f} TFLOP/s", flush=True)
    for la in (1, 0):
        h.set_option("lookahead", la); h.set_option("panel", 4); h.set_option("panel_left", -1)
        h.set_option("profile", 2); h.reset_profile(); h.loglik_batch(Th)
        print(f"  profile lookahead={la}")
        for k, v in h.profile().items():
            if v["launches"]:
                print(f"   {k:14s} {v['ms']:9.3f} ms  {int(v['launches']):5d} launches  {v['flops']/max(v['ms'],1e-9)/1e9:8.2f} TFLOP/s")
        h.set_option("profile", 0)

"""cfg 4 (200 thetas x N=4096) under option variants: evals/s and TFLOP/s per setting (developer A/B, one process)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n, B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 200
X, y = syn.make_dataset(n, 8)
Th = syn.theta_batch(B, "se_ard", 8)
Th[:, -1] = np.maximum(Th[:, -1], 0.05)
variants = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[3:]] or [{}, {"build_overlap": 0}, {}, {"build_overlap": 0}]
variants = [{k: int(v) for k, v in o.items()} for o in variants]
for opts in variants:
    h = _lib.Handle(X, y, "se_ard")
    for k, v in opts.items():
        h.set_option(k, v)
    h.loglik_batch(Th[:8]); h.loglik_batch(Th)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out, info = h.loglik_batch(Th); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[1]
    print(f"N={n} B={B} {str(opts):40s} {dt*1e3:8.2f} ms  {B/dt:8.1f} evals/s  {B*n**3/3/dt/1e12:6.2f} TFLOP/s  bad={int((info!=0).sum())}", flush=True)
    h.close()

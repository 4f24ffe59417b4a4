"""Latency outliers of the single-launch dataflow evaluation: many repetitions per (N, B), report the tail."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cases = [(1536, 1), (1536, 2), (1536, 3), (1536, 4), (1024, 4), (1024, 6), (2048, 1), (2048, 2), (1280, 3), (4096, 1), (512, 8)]
for n, B in (cases[::-1] if os.environ.get("REVERSE") else cases):
    X, y = syn.make_dataset(n, 8)
    h = _lib.Handle(X, y, "se_ard")
    for k, v in [a.split("=") for a in sys.argv[2:]]:
        h.set_option(k, int(v))
    Th = np.tile(syn.default_theta("se_ard", 8), (B, 1)) * (1 + 0.01 * np.arange(B))[:, None]
    h.loglik_batch(Th); h.loglik_batch(Th)
    ts = np.empty(reps)
    for r in range(reps):
        t0 = time.perf_counter(); h.loglik_batch(Th); ts[r] = time.perf_counter() - t0
    ts *= 1e3
    med = np.median(ts)
    out = ts[ts > 3 * med]
    print(f"N={n} B={B}: median {med:.3f} ms  p99.9 {np.quantile(ts, 0.999):.3f}  max {ts.max():.3f}  >3x median: {len(out)} {np.round(np.sort(out)[-6:], 2).tolist()}", flush=True)
    h.close()

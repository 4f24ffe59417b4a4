"""ms per single-theta evaluation across N (fp64): the mid-N latency table of DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
sizes = [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096, 8192, 16384]
for n in sizes:
    d, kernel = (1, "se") if n == 512 else (8, "se_ard")
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    h = _lib.Handle(X, y, kernel)
    h.loglik(th); h.loglik(th)
    reps = 40 if n <= 4096 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ll, info = h.loglik(th)
    dt = (time.perf_counter() - t0) / reps
    print(f"N={n} d={d}: {dt*1e3:.3f} ms/eval  {n**3/3/dt/1e12:.2f} TFLOP/s  ll={ll:.12g}", flush=True)
    h.close()

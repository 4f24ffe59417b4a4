"""A/B of the latency GEMM shape (4x4 waves, 4 LDS stages, counted waits) for launches <= latency_tiles tiles."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n, d, kernel, reps in ((512, 1, "se", 20), (4096, 8, "se_ard", 10), (8192, 8, "se_ard", 5), (32768, 8, "se_ard", 3)):
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    h = _lib.Handle(X, y, kernel)
    for lg, lt in ((0, 256), (1, 256), (1, 512), (1, 128)):
        h.set_option("latency_gemm", lg); h.set_option("latency_tiles", lt)
        h.loglik(th)
        t0 = time.perf_counter()
        for _ in range(reps):
            ll, info = h.loglik(th)
        dt = (time.perf_counter() - t0) / reps
        print(f"N={n} latency_gemm={lg} tiles<={lt}: {dt*1e3:.3f} ms/eval ll={ll:.12g}", flush=True)
    h.close()

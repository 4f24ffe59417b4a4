"""A/B of the "thin_tiles" option (gemm_nt skips the zero rows of the rhs block-row and the unread upper quadrant of
diagonal tiles): cfg 4 batch, cfg 2 multi-kernel, headline."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn

def timeit(f, reps):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps

for n, B, reps in ((4096, 200, 3), (4096, 32, 5), (2048, 200, 5), (1024, 200, 10), (32768, 1, 3)):
    X, y = syn.make_dataset(n, 8)
    Th = syn.theta_batch(max(B, 2), "se_ard", 8)[:B]
    Th[:, -1] = np.maximum(Th[:, -1], 0.05)
    if B == 1:
        Th = syn.default_theta("se_ard", 8)[None, :]
    h = _lib.Handle(X, y, "se_ard")
    for thin in (0, 1, 0, 1):
        h.set_option("thin_tiles", thin)
        dt = timeit(lambda: h.loglik_batch(Th), reps)
        print(f"N={n} B={B} thin_tiles={thin}: {dt*1e3:8.2f} ms  {B/dt:9.1f} evals/s  {B*n**3/3/dt/1e12:6.2f} TFLOP/s", flush=True)
    h.close()

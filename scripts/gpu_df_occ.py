"""64-tile dataflow kernel with ONE workgroup per CU (option dataflow_lds_kib = 84) vs two: single theta and small batches."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n in (512, 1024, 1536, 2048, 3072, 4096, 5120, 6144, 7168):
    d, kernel = 8, "se_ard"
    X, y = syn.make_dataset(n, d)
    h = _lib.Handle(X, y, kernel)
    row = [f"N={n:5d}"]
    for B in (1, 2, 4, 8):
        Th = np.tile(syn.default_theta(kernel, d), (B, 1)) * (1 + 0.01 * np.arange(B))[:, None]
        for kib in (0, 84):
            h.set_option("dataflow_lds_kib", kib)
            h.loglik_batch(Th); h.loglik_batch(Th)
            reps = 30 if n <= 4096 else 10
            t0 = time.perf_counter()
            for _ in range(reps):
                out, info = h.loglik_batch(Th)
            dt = (time.perf_counter() - t0) / reps
            row.append(f"B={B} lds={kib:2d}: {dt*1e3:7.3f} ms")
    print(" | ".join(row), flush=True)
    h.close()

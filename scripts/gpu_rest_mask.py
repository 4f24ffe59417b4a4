"""Look-ahead schedule with the trailing updates on a CU-masked stream (option rest_mask = CUs per XCD left free for the panel
stream): ms per single-theta evaluation, bit-identity vs the unmasked run.  Usage: gpu_rest_mask.py [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn

sizes = [int(a) for a in sys.argv[1:]] or [12288, 16384, 20480, 24576, 32768]
for n in sizes:
    d, kernel = 8, "se_ard"
    X, y = syn.make_dataset(n, d)
    h = _lib.Handle(X, y, kernel)
    h.set_option("rest_mask_max_nt", 1024)
    th = syn.default_theta(kernel, d)
    ref = {}
    for rnd in range(2):
        row = [f"N={n:5d}"]
        for tail in ((64, 48, 32) if n <= 16384 else (64,)):
            for force_la in ((1,) if n > 12288 else (0, 1)):
                for res in (0, 1, 2, 3, 4):
                    if not force_la and res:
                        continue
                    h.set_option("dataflow_max_nt", 64 if force_la else 96)
                    h.set_option("dataflow_tail", tail)
                    h.set_option("rest_mask", res)
                    h.loglik(th); h.loglik(th)
                    reps = 8
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        ll, info = h.loglik(th)
                    dt = (time.perf_counter() - t0) / reps
                    key = (tail, force_la)
                    ref.setdefault(key, ll)
                    flag = "" if ll == ref[key] else f" DIFF {ll - ref[key]:.2e}"
                    row.append(f"{'la' if force_la else 'df'} tail{tail} m{res}: {dt*1e3:7.3f}{flag}")
        print(" | ".join(row), flush=True)
    h.close()

"""Batches of thetas on small problems: single-launch dataflow (forced: dataflow_max_slots = 256) vs the multi-kernel
schedule (dataflow = 0) vs the default rule -- where is the crossover now that the dataflow launch picks its occupancy?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n in (256, 512, 1024, 2048):
    X, y = syn.make_dataset(n, 8)
    for B in (8, 16, 32, 64, 128, 256):
        Th = np.tile(syn.default_theta("se_ard", 8), (B, 1)) * (1 + 0.001 * np.arange(B))[:, None]
        row = [f"N={n:5d} B={B:4d} tasks={(2*((n+127)//128)+1)*(2*((n+127)//128)+2)//2*B:6d}"]
        for name, opts in (("default", {}), ("dataflow", {"dataflow_max_slots": 256}), ("multi-kernel", {"dataflow": 0})):
            h = _lib.Handle(X, y, "se_ard")
            for k, v in opts.items():
                h.set_option(k, v)
            h.loglik_batch(Th); h.loglik_batch(Th)
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                out, info = h.loglik_batch(Th)
            dt = (time.perf_counter() - t0) / reps
            row.append(f"{name}: {dt*1e3:7.3f} ms ({B/dt:8.0f}/s)")
            h.close()
        print(" | ".join(row), flush=True)

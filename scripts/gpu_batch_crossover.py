"""Batches of thetas: ONE dataflow launch for all slots against the multi-kernel batch schedule, per problem size and batch size
(round 6: the crossover sits at ~34-36 thousand 64-tile tasks for every N = 512 .. 12288 -- option dataflow_max_tasks).
   python3 scripts/gpu_batch_crossover.py [dtype]       -> profiles/r06_batch_crossover.txt (fp64), r06_batch_crossover_f32.txt"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianinference_amd import _lib, synthetic as syn
CASES = ((512, (12, 32, 64, 128, 400, 800)), (1024, (12, 24, 32, 64, 128, 200)), (1536, (12, 24, 48, 96, 128)), (2048, (12, 24, 48, 64, 96)),
         (3072, (8, 16, 24, 32, 48)), (4096, (4, 8, 12, 16, 24)), (6144, (2, 4, 6, 8, 12)), (8192, (2, 3, 4, 6, 8)), (12288, (2, 3, 4)))
dtype = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for n, Bs in CASES:
    d, kern = 8, "se_ard"
    X, y = syn.make_dataset(n, d)
    Th = syn.theta_batch(max(Bs), kern, d); Th[:, -1] = np.maximum(Th[:, -1], 0.05)
    h = _lib.Handle(X, y, kern, dtype=dtype)
    for B in Bs:
        row = []
        for opts in ({"dataflow_max_slots": 1}, {"dataflow_max_slots": 100000}, {"dataflow_max_slots": -1}):
            for k, v in opts.items():
                h.set_option(k, v)
            h.loglik_batch(Th[:B]); h.loglik_batch(Th[:B])
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); _, info = h.loglik_batch(Th[:B]); ts.append(time.perf_counter() - t0)
            row.append(float(np.median(ts)) * 1e3)
        nt = n // 128 * (2 if dtype == 64 else 1)                 # fp32 runs 128-tiles
        tasks = (nt + 1) * (nt + 2) // 2 * B
        print(f"N={n:5d} batch {B:3d} ({tasks:6d} tasks): multi-kernel {row[0]:8.3f} ms | one dataflow launch {row[1]:8.3f} ms | "
              f"library's choice {row[2]:8.3f} ms = {B / row[2] * 1e3:8.0f} evals/s", flush=True)
    h.close()

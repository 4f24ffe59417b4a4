"""Longer soak of the dataflow schedule: random sizes / kernels / batch sizes, every result checked against the
multi-kernel schedule (1e-9 relative) and for bit-repeatability.  Prints a summary line."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bayesianinference_amd import _lib, synthetic as syn
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t_end = time.time() + budget
cases = evals = bad = 0
while time.time() < t_end:
    n = int(rng.choice([70, 129, 300, 640, 1000, 1537, 2048, 3000, 4096, 5000, 6200, 7000, 8200, 9100]))
    d = int(rng.choice([1, 2, 3, 8]))
    kernel = str(rng.choice(["se_ard", "matern52_ard"]))
    B = int(rng.choice([1, 1, 1, 2, 5, 8, 16]))
    X, y = syn.make_dataset(n, d, seed=int(rng.integers(1 << 30)))
    base = syn.default_theta(kernel, d)
    Th = np.stack([base * (1 + 0.2 * rng.random(len(base))) for _ in range(B)])
    h = _lib.Handle(X, y, kernel)
    h.set_option("dataflow", 0)
    ref, iref = h.loglik_batch(Th)
    h.set_option("dataflow", 1)
    h.set_option("dataflow_lds_kib", int(rng.choice([-1, -1, 0, 84])))      # occupancy rule: auto / two per CU / one per CU
    h.set_option("dataflow_occ3", int(rng.choice([-1, -1, 0, 1])))          # three-per-CU build: auto / never / always
    first = None
    for rep in range(int(rng.integers(3, 12))):
        out, info = h.loglik_batch(Th)
        evals += B
        ok = np.array_equal(info, iref) and np.allclose(out, ref, rtol=1e-9, atol=1e-9 * n)
        if first is None:
            first = out.copy()
        ok = ok and np.array_equal(out, first)
        bad += not ok
    h.close()
    cases += 1
print(f"soak: {cases} cases, {evals} evaluations, {bad} failures", flush=True)

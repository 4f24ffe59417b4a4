"""Dataflow Cholesky vs the multi-kernel schedule: values and time (developer check; parity against the
oracle is the job of tests/)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bayesianinference_amd import _lib, synthetic as syn
for n, d, kernel in ((200, 2, "se_ard"), (512, 1, "se"), (1000, 3, "matern52_ard"), (4096, 8, "se_ard"), (8192, 8, "se_ard")):
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    h = _lib.Handle(X, y, kernel)
    res = {}
    for df in (0, 1, 2):
        h.set_option("dataflow", 1 if df else 0)
        h.set_option("dataflow_fine_nt", 16 if df == 2 else 0)
        ll, ld, qd, info = h.loglik_parts(th)
        reps = 20 if n <= 4096 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            h.loglik(th)
        res[df] = (ll, ld, qd, info, (time.perf_counter() - t0) / reps)
    print(f"N={n}: multi-kernel {res[0][4]*1e3:.3f} ms  dataflow128 {res[1][4]*1e3:.3f} ms  dataflow64 {res[2][4]*1e3:.3f} ms | ll {res[0][0]:.12g} / {res[1][0]:.12g} / {res[2][0]:.12g} | info {res[0][3]} {res[1][3]} {res[2][3]}", flush=True)
    if n <= 1000:
        Th = np.stack([th * (1 + 0.05 * k) for k in range(7)])
        h.set_option("dataflow", 0); a, ia = h.loglik_batch(Th)
        h.set_option("dataflow", 1); b, ib = h.loglik_batch(Th)
        assert h.fit(th) == 0
        pm, pv = h.predict(syn.make_test_points(20, d))
        h.set_option("dataflow", 0)
        assert h.fit(th) == 0
        qm, qv = h.predict(syn.make_test_points(20, d))
        print("   predict after fine fit vs multi-kernel: max rel", float(np.max(np.abs(pm - qm) / (np.abs(qm) + 1e-300))), float(np.max(np.abs(pv - qv) / np.abs(qv))))
        print("   batch max rel diff", float(np.max(np.abs(a - b) / np.abs(a))), ia.tolist(), ib.tolist(), flush=True)
    h.close()

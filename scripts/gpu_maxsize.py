"""Largest-problem check (one MI355X, 288 GB): fp64 N=131072 (137 GB factor) and a ragged N=160000
(205 GB), SE-ARD d=8.  No oracle is affordable at these sizes, so the same size-independent
properties as tests/test_gpu_parity.py::test_full_size_properties_n32768 are checked: the bordered
quadratic form equals y^T (K^-1 y) from an independent forward+backward solve, and sampled rows of
K (K^-1 y) reproduce y.  Prints one JSON line per size."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bayesianinference_amd import _lib, synthetic as syn      # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [131072, 160000]
d = 8
for n in sizes:
    rec = {"N": n, "d": d, "dtype": "f64", "factor_GB": round(8e-9 * n * n / 2, 1)}    # packed lower triangle
    try:
        X, y = syn.make_dataset(n, d)
        th = syn.default_theta("se_ard", d)
        h = _lib.Handle(X, y, "se_ard")
        t0 = time.perf_counter()
        ll, ld, qd, info = h.loglik_parts(th)
        rec["first_eval_s"] = round(time.perf_counter() - t0, 3)
        t0 = time.perf_counter()
        ll2, ld2, qd2, info2 = h.loglik_parts(th)
        dt = time.perf_counter() - t0
        rec.update(eval_s=round(dt, 3), cholesky_tflops=round(n ** 3 / 3 / dt / 1e12, 2), info=int(info),
                   loglik=ll, logdet=ld, quad=qd, repeatable=bool(ll2 == ll and ld2 == ld))
        assert h.fit(th) == 0
        t0 = time.perf_counter()
        alpha = h.solve(y)
        rec["solve_s"] = round(time.perf_counter() - t0, 3)
        rec["quad_vs_solve_rel"] = float(abs(y @ alpha - qd) / abs(qd))
        idx = np.array([0, 1, 4097, n // 2 + 3, n - 1])
        ell, sf, sn = th[:d], th[d], th[d + 1]                  # SE-ARD rows of K, written out here (numpy)
        r2 = (((X[idx][:, None, :] - X[None, :, :]) / ell) ** 2).sum(axis=2)
        Krows = sf * sf * np.exp(-0.5 * r2)
        Krows[np.arange(len(idx)), idx] += sn * sn
        rec["residual_max_abs"] = float(np.abs(Krows @ alpha - y[idx]).max())
        h.close()
    except Exception as e:                                     # e.g. out of device memory: report, go on
        rec["error"] = repr(e)[:300]
    print(json.dumps(rec), flush=True)

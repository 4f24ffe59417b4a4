"""gphip_predict after a single-launch fit: forward substitution as one dataflow launch (df) against the multi-kernel substitution
(mk), by number of test points.   python scripts/gpu_predict_df_sweep.py"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from bayesianinference_amd import _lib, synthetic as syn
for n in (2048, 4096, 8192, 12288):
    X, y = syn.make_dataset(n, 8); th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    h.fit(th)
    row = [f"N={n}"]
    for m in (100, 512, 1024, 2048, 4096, 8192):
        if m > n: continue
        Xs = syn.make_test_points(m, 8)
        r = {}
        for mode in (1, 0):
            h.set_option("predict_df", (1 << 20) if mode else 0)
            h.predict(Xs)
            t0 = time.perf_counter()
            for _ in range(5): h.predict(Xs)
            r[mode] = (time.perf_counter() - t0) / 5 * 1e3
        row.append(f"M={m}: df {r[1]:.2f} / mk {r[0]:.2f}")
    print(" | ".join(row), flush=True)
    h.close()

"""gphip_fit and gphip_predict next to gphip_loglik per size, with the per-class launch counts of one fit and one prediction.
   python scripts/gpu_fit_time.py [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
sizes = [int(a) for a in sys.argv[1:]] or [512, 2048, 4096, 8192, 16384]
for n in sizes:
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    Xs = syn.make_test_points(100, 8)
    h = _lib.Handle(X, y, "se_ard")
    h.loglik(th); h.fit(th); h.predict(Xs)
    def t(f, reps=5):
        t0 = time.perf_counter()
        for _ in range(reps): f()
        return (time.perf_counter() - t0) / reps * 1e3
    tl, tf, tp = t(lambda: h.loglik(th)), t(lambda: h.fit(th)), t(lambda: h.predict(Xs))
    print(f"N={n}: loglik {tl:.3f} ms | fit {tf:.3f} ms | predict 100 points {tp:.3f} ms", flush=True)
    for name, f in (("fit", lambda: h.fit(th)), ("predict", lambda: h.predict(Xs))):
        h.set_option("profile", 2); h.reset_profile(); f()
        print("    " + name + ": " + ", ".join(f"{k} {v['ms']:.3f} ms/{int(v['launches'])}" for k, v in h.profile().items() if v["launches"]))
        h.set_option("profile", 0)
    h.close()

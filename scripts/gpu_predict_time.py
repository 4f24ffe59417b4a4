import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from bayesianinference_amd import _lib, synthetic as syn
for n in (8192, 12288, 16384, 32768):
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    assert h.fit(th) == 0
    for m in (100, 2000):
        Xs = syn.make_test_points(m, 8)
        h.predict(Xs)
        h.set_option("profile", 2); h.reset_profile()
        t0 = time.perf_counter()
        for _ in range(3):
            mu, var = h.predict(Xs)
        dt = (time.perf_counter() - t0) / 3
        pr = h.profile(); h.set_option("profile", 0)
        kms = sum(v["ms"] for v in pr.values()) / 3
        nl = sum(v["launches"] for v in pr.values()) / 3
        print(f"N={n} M={m}: predict {dt*1e3:.2f} ms/call; kernels {kms:.2f} ms in {nl:.0f} launches", flush=True)
    h.close()

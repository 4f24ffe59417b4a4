"""Developer probe for the GPU box: step-wise parity diffs and a per-kernel-class timing table."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402


def parity():
    for kernel, d, n in (("se", 1, 200), ("se_ard", 8, 300), ("matern52_ard", 5, 129)):
        X, y = syn.make_dataset(n, d)
        th = syn.theta_batch(1, kernel, d)[0]
        h = _lib.Handle(X, y, kernel)
        K = h.covariance(th)
        Ko = orc.covariance_matrix(kernel, th, X)
        print(f"cov {kernel} n={n}: max rel err {np.max(np.abs(K - Ko) / np.abs(Ko)):.3e}")
        th[-1] = max(th[-1], 0.05)
        got = h.loglik_parts(th)
        want = orc.log_likelihood(kernel, th, X, y, parts=True)
        print(f"  loglik got {got} want {want}")
        if h.fit(th) == 0:
            Xs = syn.make_test_points(7, d)
            mu, var = h.predict(Xs)
            mo, so = orc.predict_internal(kernel, th, X, y, Xs)
            print(f"  predict mu err {np.max(np.abs(mu - mo)):.3e} sd err {np.max(np.abs(np.sqrt(var) - so)):.3e}")
        h.close()


def timing(ns, panel=4, reps=3):
    for n in ns:
        X, y = syn.make_dataset(n, 8)
        th = syn.default_theta("se_ard", 8)
        h = _lib.Handle(X, y, "se_ard")
        h.set_option("panel", panel)
        h.loglik(th)
        t0 = time.perf_counter()
        for _ in range(reps):
            ll, info = h.loglik(th)
        dt = (time.perf_counter() - t0) / reps
        print(f"N={n} panel={panel}: {dt*1e3:.2f} ms/eval  loglik={ll:.10g} info={info}  "
              f"chol {n**3/3/dt/1e12:.2f} TFLOP/s")
        h.set_option("profile", 2)
        h.reset_profile()
        h.loglik(th)
        for k, v in h.profile().items():
            if v["launches"]:
                extra = ""
                if v["flops"]:
                    extra += f" {v['flops']/v['ms']/1e9:.2f} TFLOP/s"
                if v["bytes"]:
                    extra += f" {v['bytes']/v['ms']/1e6:.1f} GB/s"
                print(f"   {k:14s} {v['ms']:9.3f} ms  {int(v['launches']):5d} launches{extra}")
        h.close()


if __name__ == "__main__":
    print(_lib.load().gphip_version().decode(), "devices:", _lib.device_count())
    if "--parity" in sys.argv or len(sys.argv) == 1:
        parity()
    if "--time" in sys.argv or len(sys.argv) == 1:
        timing([2048, 8192])
    if "--big" in sys.argv:
        timing([32768], reps=2)

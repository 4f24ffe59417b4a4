"""Kernel-build time of a run-time compiled covariance function (SE-ARD given as source text) next to the named SE-ARD kernel,
HIP events around the build launches (profile = 2), and the whole evaluation.   python scripts/gpu_custom_kbuild.py [N [d]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
BODY = "T s = 0; for (int k = 0; k < D; ++k) { const T u = (X(k) - Y(k)) / P(k); s += u * u; } return P(D) * P(D) * exp((T)-0.5 * s);"
X, y = syn.make_dataset(n, d)
th = syn.default_theta("se_ard", d)
# the same function as Mathematica's CForm prints it (Power with integer exponents, one quotient per dimension)
CFORM = "return Power(P(%d),2)*Exp(-0.5*(%s));" % (d, " + ".join(f"Power(X({k}) - Y({k}),2)/Power(P({k}),2)" for k in range(d)))
for name, kern in (("named se_ard", "se_ard"), ("source text", _lib.CustomKernel(BODY, d + 1)), ("CForm text", _lib.CustomKernel(CFORM, d + 1))):
    t0 = time.perf_counter()
    h = _lib.Handle(X, y, kern)
    tc = time.perf_counter() - t0
    h.loglik(th); h.loglik(th)
    t0 = time.perf_counter()
    for _ in range(5):
        r = h.loglik(th)
    wall = (time.perf_counter() - t0) / 5 * 1e3
    h.set_option("profile", 2); h.loglik(th); h.reset_profile(); h.loglik(th)
    kb = h.profile()["kbuild"]
    print(f"N={n} d={d} {name:13s}: create {tc*1e3:7.1f} ms | evaluation {wall:8.2f} ms | kernel build {kb['ms']:.3f} ms "
          f"({8.0 * (n * (n + 1) / 2 + n * d) / kb['ms'] / 1e6 / 8000:.2f} of 8 TB/s) | ll {r[0]:.10g}", flush=True)
    h.close()

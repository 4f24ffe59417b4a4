"""Kernel-build A/B of two library builds on one box: GPHIP_LIB=<variant> python scripts/gpu_kbuild_ab.py [tag]
fp64 SE-ARD N=32768 d=8 (the BASELINE metric's build) and fp32 Matern-5/2 N=32768 d=16: HIP-event time of the kbuild
launches inside likelihood evaluations (profile class "kbuild"), algorithmic bytes / time vs 8 TB/s; plus the covariance
matrix itself at N=700 (saved so the two builds can be compared entry by entry)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
tag = sys.argv[1] if len(sys.argv) > 1 else "x"
for n, d, kernel, dtype, reps in ((32768, 8, "se_ard", 64, 6), (32768, 16, "matern52_ard", 32, 4), (32768, 8, "matern52_ard", 64, 3),
                                  (32768, 16, "se_ard", 32, 3)):
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d, dtype="f32" if dtype == 32 else "f64")
    h = _lib.Handle(X, y, kernel, dtype=dtype)
    h.set_option("profile", 1)
    h.loglik(th)
    h.reset_profile()
    for i in range(reps):
        h.loglik(th * (1 + 0.001 * i))
    p = h.profile()["kbuild"]
    nl = int(p["launches"])
    ms = p["ms"] / max(nl, 1)
    es = 8 if dtype == 64 else 4
    gb = es * (n * (n + 1) / 2 + n * d) / 1e9
    print(f"{tag}: kbuild {kernel} fp{dtype} N={n} d={d}: {ms:.3f} ms/launch over {nl} launches = {gb / ms:.2f} TB/s = {gb / ms / 8.0:.3f} of 8 TB/s", flush=True)
    h.close()
for kernel, dtype, d in (("se_ard", 64, 8), ("matern52_ard", 64, 8), ("matern52_ard", 32, 16), ("se_ard", 32, 16), ("se", 64, 1)):
    X, y = syn.make_dataset(700, d)
    th = syn.default_theta(kernel, d)
    h = _lib.Handle(X, y, kernel, dtype=dtype)
    K = h.covariance(th)
    Xs = syn.make_test_points(300, d)
    k, kappa = h.cross_covariance(th, Xs)
    np.savez(os.path.join(ROOT, "gpurun_out", f"kab_{tag}_{kernel}_{dtype}.npz"), K=K, k=k)
    h.close()

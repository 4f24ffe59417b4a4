"""BASELINE.json config 5 (single-GPU part): Matern-5/2 ARD, N=65536 d=16 fp32, fit + prediction on 10k
test points; plus an fp64 regression check of the headline config."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
d = 16
X, y = syn.make_dataset(n, d)
th = syn.default_theta("matern52_ard", d, dtype="f32")
h = _lib.Handle(X, y, "matern52_ard", dtype=32)
h.loglik(th)
t0 = time.perf_counter(); ll, info = h.loglik(th); dt = time.perf_counter() - t0
print(f"fp32 N={n} d={d}: loglik {dt*1e3:.1f} ms ({n**3/3/dt/1e12:.1f} TFLOP/s of 157.3 fp32 MFMA) ll={ll:.6g} info={info}", flush=True)
h.set_option("profile", 2); h.reset_profile(); h.loglik(th)
for k, v in h.profile().items():
    if v["launches"]:
        extra = f" {v['flops']/v['ms']/1e9:.1f} TFLOP/s" if v["flops"] else ""
        extra += f" {v['bytes']/v['ms']/1e6:.0f} GB/s" if v["bytes"] else ""
        print(f"   {k:14s} {v['ms']:9.3f} ms {int(v['launches']):5d} launches{extra}")
h.set_option("profile", 0)
t0 = time.perf_counter(); assert h.fit(th) == 0; tf = time.perf_counter() - t0
Xs = syn.make_test_points(m, d)
t0 = time.perf_counter(); mu, var = h.predict(Xs); tp = time.perf_counter() - t0
print(f"fit {tf*1e3:.1f} ms; predict M={m}: {tp*1e3:.1f} ms ({n*n*m/tp/1e12:.1f} TFLOP/s on the N^2 M solve), "
      f"mean range [{mu.min():.3f}, {mu.max():.3f}], var range [{var.min():.4f}, {var.max():.4f}]", flush=True)
mu_tr, var_tr = h.predict(X[:256])
print(f"prediction at training inputs: rms(mu - y) = {np.sqrt(np.mean((mu_tr - y[:256])**2)):.4f} (noise sd 0.1, sn=0.3)")
h.close()

"""Cost of gphip_loglik_grad next to gphip_loglik, per size, with the per-class HIP-event profile of one gradient call.
   python scripts/gpu_grad_time.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768]
for n in sizes:
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    h.loglik_grad(th)
    t0 = time.perf_counter(); ll, g, info = h.loglik_grad(th); dt = time.perf_counter() - t0
    h.loglik(th)
    t0 = time.perf_counter(); h.loglik(th); dl = time.perf_counter() - t0
    # K^-1 from the factor costs 2 N^3 / 3 on top of the factorisation's N^3 / 3
    print(f"N={n}: loglik {dl*1e3:.1f} ms, loglik+grad {dt*1e3:.1f} ms (x{dt/dl:.2f}) = {n**3 / dt / 1e12:.1f} TFLOP/s over N^3; |grad|max={abs(g).max():.3g}", flush=True)
    h.set_option("profile", 2); h.reset_profile(); h.loglik_grad(th)
    for k, v in h.profile().items():
        if v["launches"]:
            print(f"     {k:16s} {v['ms']:9.3f} ms  {int(v['launches']):5d} launches")
    h.close()

"""Single-theta latency at small N (the reference's sequential-chain usage) + per-class profile."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
sizes = [int(a) for a in sys.argv[1:]]
for n, d, kernel in ([(n_, 8, "se_ard") for n_ in sizes] or [(512, 1, "se"), (4096, 8, "se_ard")]):
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    h = _lib.Handle(X, y, kernel)
    h.loglik(th)
    t0 = time.perf_counter()
    for _ in range(20):
        ll, info = h.loglik(th)
    dt = (time.perf_counter() - t0) / 20
    print(f"N={n} d={d}: {dt*1e3:.3f} ms/eval ({1/dt:.0f} evals/s) ll={ll:.12g}", flush=True)
    h.set_option("profile", 2); h.reset_profile(); h.loglik(th)
    for k, v in h.profile().items():
        if v["launches"]:
            print(f"   {k:14s} {v['ms']:9.3f} ms {int(v['launches']):5d} launches ({v['ms']/v['launches']*1e3:.1f} us each)")
    h.close()

"""gphip_solve after a fit: both substitutions as dataflow launches (option predict_df > 0) against the multi-kernel substitutions.
   python scripts/gpu_solve_time.py N nrhs"""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from bayesianinference_amd import _lib, synthetic as syn
n, m = int(sys.argv[1]), int(sys.argv[2])
X, y = syn.make_dataset(n, 8); th = syn.default_theta("se_ard", 8)
h = _lib.Handle(X, y, "se_ard"); h.fit(th)
b = np.random.default_rng(0).standard_normal((n, m)) if m > 1 else np.random.default_rng(0).standard_normal(n)
r = {}
for mode in (0, 2048):
    h.set_option("predict_df", mode); h.fit(th)
    t0 = time.perf_counter(); x = h.solve(b); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); x = h.solve(b); t2 = time.perf_counter() - t0
    r[mode] = x
    print(f"N={n} nrhs={m} predict_df={mode}: first {t1*1e3:.2f} ms, second {t2*1e3:.2f} ms", flush=True)
print("max rel diff", np.abs(r[2048]-r[0]).max()/np.abs(r[0]).max(), flush=True)

"""Does the 64-tile dataflow schedule still win above N=8192?  (developer check)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n in (8192, 10240, 12288, 16384, 20480):
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    res = []
    for max_nt in (64, 256):
        h.set_option("dataflow_max_nt", max_nt); h.set_option("dataflow_fine_nt", max_nt)
        h.loglik(th); h.loglik(th)
        t0 = time.perf_counter()
        for _ in range(5):
            ll, info = h.loglik(th)
        res.append(((time.perf_counter() - t0) / 5, ll))
    print(f"N={n}: look-ahead+tail {res[0][0]*1e3:.2f} ms   all-dataflow64 {res[1][0]*1e3:.2f} ms   rel diff {abs(res[0][1]-res[1][1])/abs(res[0][1]):.1e}", flush=True)
    h.close()

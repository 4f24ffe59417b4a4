"""Where does the single-launch dataflow schedule stop winning?  ms per evaluation for N = 12288 .. 18432 with the
dataflow range extended (dataflow_max_nt) vs the look-ahead schedule with dataflow tail."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n in (12288, 13312, 14336, 16384, 18432):
    X, y = syn.make_dataset(n, 8); th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard"); h.loglik(th)
    row = []
    for mx in (96, 160, 96, 160):
        h.set_option("dataflow_max_nt", mx); h.set_option("dataflow_fine_nt", mx); h.loglik(th)
        t0 = time.perf_counter()
        for _ in range(4): ll, info = h.loglik(th)
        row.append(f"max_nt={mx}: {(time.perf_counter()-t0)/4*1e3:6.2f}")
    print(f"N={n}: " + "  ".join(row) + f"  ll={ll:.8g}", flush=True)
    h.close()

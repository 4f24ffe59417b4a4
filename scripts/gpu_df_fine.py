"""Where do 64x64 dataflow tiles stop paying?  ms/eval for fine_nt on/off across N (developer check)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
for n in (256, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192):
    d, kernel = (1, "se") if n == 512 else (8, "se_ard")
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    h = _lib.Handle(X, y, kernel)
    out = []
    for fine in (0, 64):
        h.set_option("dataflow_fine_nt", fine)
        h.loglik(th); h.loglik(th)
        reps = 30 if n <= 2048 else 8
        t0 = time.perf_counter()
        for _ in range(reps):
            ll, info = h.loglik(th)
        out.append(((time.perf_counter() - t0) / reps, ll))
    print(f"N={n}: tiles128 {out[0][0]*1e3:.3f} ms  tiles64 {out[1][0]*1e3:.3f} ms  rel diff {abs(out[0][1]-out[1][1])/abs(out[0][1]):.1e}", flush=True)
    h.close()

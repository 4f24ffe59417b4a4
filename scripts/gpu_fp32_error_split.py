"""Where the fp32 path's error sits (cfg 5: Matern-5/2, d = 16): log det and quadratic form of the fp32 evaluation against the
fp64 evaluation of the same data (itself at 1e-8 of the oracle, tests/test_gpu_configs.py), per size.  Decides what a fp64
refinement step of alpha could buy: it corrects the quadratic form only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianinference_amd import _lib, synthetic as syn
d = 16
for n in [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768, 65536]:
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("matern52_ard", d, dtype="f32")
    res = {}
    for dt in (32, 64):
        h = _lib.Handle(X, y, "matern52_ard", dtype=dt)
        res[dt] = h.loglik_parts(th)
        h.close()
    (l32, d32, q32, _), (l64, d64, q64, _) = res[32], res[64]
    print(f"N={n}: loglik {l64:.6f}  rel err fp32 {abs(l32 - l64) / abs(l64):.2e} | log det {d64:.4f}: abs err {d32 - d64:+.4f} (rel {abs(d32 - d64) / abs(d64):.2e}) | "
          f"quad {q64:.4f}: abs err {q32 - q64:+.4f} (rel {abs(q32 - q64) / abs(q64):.2e})", flush=True)

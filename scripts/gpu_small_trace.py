import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = 4096
X, y = syn.make_dataset(n, 8)
th = syn.default_theta("se_ard", 8)
h = _lib.Handle(X, y, "se_ard")
h.set_option("latency_gemm", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for _ in range(6):
    h.loglik(th)
h.close()

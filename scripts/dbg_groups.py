import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from bayesianinference_amd import _lib, synthetic as syn
n, B = 4096, int(sys.argv[1]); G = int(sys.argv[2])
X, y = syn.make_dataset(n, 8)
Th = syn.theta_batch(B, "se_ard", 8); Th[:, -1] = np.maximum(Th[:, -1], 0.05)
h = _lib.Handle(X, y, "se_ard")
h.set_option("batch_group_min", 8)
h.set_option("batch_groups", 1)
ref, info0 = h.loglik_batch(Th)
h.set_option("batch_groups", G)
for i in range(3):
    try:
        out, info = h.loglik_batch(Th)
        print("G", G, "B", B, "try", i, "equal", np.array_equal(out, ref), np.array_equal(info, info0), flush=True)
    except Exception as e:
        print("ERR", e, flush=True)

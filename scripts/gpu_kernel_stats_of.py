"""rocprofv3 --kernel-trace --stats target: ONE API call pattern repeated a few times, chosen on the command line, so that
`scripts/kernel_stats_of.sh <what> <N>` prints where a call's device time goes (how the atomics of the gradient reduction
were found in round 6).   python3 scripts/gpu_kernel_stats_of.py {grad|predict100|predict1000|solve1|solve24|samples} N [dtype]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianinference_amd import _lib, synthetic as syn
what, n = sys.argv[1], int(sys.argv[2])
dtype = int(sys.argv[3]) if len(sys.argv) > 3 else 64
d = 8
X, y = syn.make_dataset(n, d)
th = syn.default_theta("se_ard", d)
h = _lib.Handle(X, y, "se_ard", dtype=dtype)
Xs = syn.make_test_points(1000 if what == "predict1000" else 100, d)
NS = int(os.environ.get("KSO_SAMPLES", "4"))
S = np.tile(th, (NS, 1)) * (1 + 0.05 * np.random.default_rng(0).random((NS, len(th))))
h.fit(th)
f = {"grad": lambda: h.loglik_grad(th), "predict100": lambda: h.predict(Xs), "predict1000": lambda: h.predict(Xs),
     "solve1": lambda: h.solve(y), "solve24": lambda: h.solve(np.tile(y[:, None], (1, 24))),
     "samples": lambda: h.predict_samples(S, Xs)}[what]
for _ in range(4):
    f()

import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
LOG2PI = np.log(2 * np.pi)
def ref(K, r):
    L = np.linalg.cholesky(K); z = np.linalg.solve(L, r)
    return -0.5 * (len(r) * LOG2PI + 2 * np.log(np.diag(L)).sum() + float(z @ z))
n, d, kernel, mean = 777, 1, "matern52", "const"
X, y = syn.make_dataset(n, d, seed=123)
rng = np.random.default_rng(5)
def rand_theta():
    th = syn.default_theta(kernel, d) * (0.6 + 0.8 * rng.random(3)); th[-1] = 0.1 + 0.4 * rng.random()
    return np.append(th, rng.normal(0, 0.3))
def check(h, tag, B):
    Th = np.stack([rand_theta() for _ in range(B)])
    out, info = h.loglik_batch(Th)
    bad = []
    for b in range(B):
        w = ref(h.covariance(Th[b]), y - Th[b][-1])
        if info[b] != 0 or abs(out[b] - w) > 1e-8 * max(abs(w), n): bad.append((b, out[b], w))
    print(tag, "B=%d" % B, "OK" if not bad else bad[:3], flush=True)
for seq in (["batch10"], ["parts", "batch10"], ["samples3", "batch10"], ["samples3", "parts", "batch10"], ["cross", "samples5", "samples2", "parts", "batch10"],
            ["batch9"], ["batch10", "batch10"], ["batch12"], ["samples3", "batch12"]):
    h = _lib.Handle(X, y, kernel, mean)
    for op in seq:
        if op.startswith("batch"): check(h, "+".join(seq), int(op[5:]))
        elif op == "parts": h.loglik_parts(rand_theta())
        elif op.startswith("samples"):
            S = int(op[7:]); h.predict_samples(np.stack([rand_theta() for _ in range(S)]), syn.make_test_points(40, d))
        elif op == "cross": h.cross_covariance(rand_theta(), syn.make_test_points(9, d))
    h.close()

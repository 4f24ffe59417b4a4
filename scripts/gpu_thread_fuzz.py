"""Three host threads, each with its own handles, hammering the C ABI concurrently (SURVEY.md 8b threading contract:
different handles may be used concurrently).  Values checked against numpy on the library's own covariance."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
LOG2PI = np.log(2 * np.pi)
errors, counts = [], [0, 0, 0]

def ref(K, r):
    L = np.linalg.cholesky(K); z = np.linalg.solve(L, r)
    return -0.5 * (len(r) * LOG2PI + 2 * np.log(np.diag(L)).sum() + float(z @ z))

def worker(tid):
    rng = np.random.default_rng(100 + tid)
    t_end = time.time() + budget
    try:
        while time.time() < t_end:
            n = int(rng.choice([129, 300, 777, 1100, 2500])); d = int(rng.choice([1, 3, 8]))
            kernel = str(rng.choice(["se_ard", "matern52_ard"]))
            world = int(rng.choice([1, 1, 2]))
            X, y = syn.make_dataset(n, d, seed=int(rng.integers(1 << 30)))
            h = _lib.Handle(X, y, kernel, device=([0] * world if world > 1 else None))
            if world > 1: h.set_option("shard_min_n", int(rng.choice([0, 1 << 30])))
            base = syn.default_theta(kernel, d)
            for _ in range(int(rng.integers(3, 9))):
                B = int(rng.choice([1, 1, 3, 12]))
                Th = np.stack([base * (0.7 + 0.6 * rng.random(len(base))) for _ in range(B)])
                out, info = h.loglik_batch(Th)
                b = int(rng.integers(B))
                w = ref(h.covariance(Th[b]), y)
                if info[b] != 0 or abs(out[b] - w) > 1e-8 * max(abs(w), n):
                    errors.append((tid, n, kernel, world, B, b, out[b], w, int(info[b])))
                if rng.random() < 0.4:
                    assert h.fit(Th[b]) == 0
                    mu, var = h.predict(syn.make_test_points(int(rng.choice([3, 600])), d))
                    if not (np.all(np.isfinite(mu)) and np.all(var > 0)): errors.append((tid, "predict", n))
                counts[tid] += 1
            h.close()
    except Exception as exc:                                   # noqa: BLE001
        errors.append((tid, repr(exc)))

ts = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
[t.start() for t in ts]; [t.join() for t in ts]
print(f"thread fuzz: calls per thread {counts}, errors: {errors[:5] if errors else 0}", flush=True)

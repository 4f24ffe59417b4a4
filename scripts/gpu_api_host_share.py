"""Round 6 (how the launch-bound substitutions after a look-ahead-schedule fit were found): wall time per call next to the HIP-event time of its launches (profile = 2), for the API calls around the likelihood: finds calls
whose cost is host-side (many small launches, host transfers) rather than device work."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from bayesianinference_amd import _lib, synthetic as syn
for n in ([int(a) for a in sys.argv[1:]] or [2048, 16384]):
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    Xs = syn.make_test_points(100, 8)
    Xs1k = syn.make_test_points(1000, 8)
    S = np.tile(th, (8, 1)) * (1 + 0.05 * np.random.default_rng(0).random((8, len(th))))
    calls = {"loglik": lambda: h.loglik(th), "fit": lambda: h.fit(th), "fit+predict100": lambda: (h.fit(th), h.predict(Xs)),
             "grad": lambda: h.loglik_grad(th), "batch8": lambda: h.loglik_batch(S),
             "fit+cross100": lambda: (h.fit(th), h.cross_covariance(th, Xs)), "fit+logdet": lambda: (h.fit(th), h.logdet()),
             "fit+solve1": lambda: (h.fit(th), h.solve(y)), "fit+predict1000": lambda: (h.fit(th), h.predict(Xs1k)),
             "predict_samples 4x100": lambda: h.predict_samples(S[:4], Xs), "fit+solve24": lambda: (h.fit(th), h.solve(np.tile(y[:, None], (1, 24))))}
    for name, f in calls.items():
        try:
            f(); 
            h.set_option("profile", 2); h.reset_profile()
            t0 = time.perf_counter()
            for _ in range(3):
                f()
            dt = (time.perf_counter() - t0) / 3
            pr = h.profile(); h.set_option("profile", 0)
            kms = sum(v["ms"] for v in pr.values() if v is not pr.get("eval_total")) / 3
            nl = sum(v["launches"] for k, v in pr.items() if k != "eval_total") / 3
            print(f"N={n} {name:22s}: {dt*1e3:8.2f} ms/call; launches {kms:8.2f} ms in {nl:5.0f}  -> host share {(dt*1e3-kms)/(dt*1e3)*100:5.1f} %", flush=True)
        except Exception as e:
            print(f"N={n} {name}: {e!r}"[:150], flush=True)
    h.close()

"""Effect of reserving CUs for the panel stream (GPHIP_RESERVE_CUS) on one evaluation."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from bayesianinference_amd import _lib, synthetic as syn
    n = int(sys.argv[2])
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    h.loglik(th)
    reps = 10 if n <= 8192 else 3
    t0 = time.perf_counter()
    for _ in range(reps):
        ll, info = h.loglik(th)
    print(f"N={n} reserve={os.environ.get('GPHIP_RESERVE_CUS')}: {(time.perf_counter()-t0)/reps*1e3:.3f} ms/eval ll={ll:.12g}", flush=True)
else:
    for n in (2048, 4096, 8192, 16384, 32768):
        for r in ("0", "8", "16", "32"):
            env = dict(os.environ, GPHIP_RESERVE_CUS=r)
            out = subprocess.run([sys.executable, __file__, "child", str(n)], env=env, capture_output=True, text=True)
            print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)

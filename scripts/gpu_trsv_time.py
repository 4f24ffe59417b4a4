"""gphip_solve of 1 .. 4 vectors: the single-launch-per-triangle substitution (gp_trsv.h) against the GEMM-shaped one, per size.
Times are host wall-clock per call (upload + 2 launches + download) and, from the library's profile class 2, the launches alone.
Output -> profiles/r06_trsv_time.txt"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianinference_amd import _lib, synthetic as syn

sizes = [int(a) for a in sys.argv[1:] if not a.startswith("m")] or [2048, 4096, 8192, 16384, 32768]
modes = [int(a[1:]) for a in sys.argv[1:] if a.startswith("m")] or [1, 0]
for n in sizes:
    d = 8
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d)
    h = _lib.Handle(X, y, "se_ard")
    assert h.fit(th) == 0
    rng = np.random.default_rng(1)
    line = f"N={n:6d}"
    for nrhs in (1, 4):
        B = rng.standard_normal((n, nrhs))
        for trsv in modes:
            h.set_option("trsv", trsv)
            for _ in range(3):
                x = h.solve(B)
            h.set_option("profile", 2); h.reset_profile()
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                x = h.solve(B)
            dt = (time.perf_counter() - t0) / reps * 1e3
            pr = h.profile()
            h.set_option("profile", 0)
            kms = pr[_lib.PROFILE_CLASSES[2]]["ms"] / reps
            hbm = 2 * n * n / 2 * 8 / 1e9
            line += f" | nrhs={nrhs} trsv={trsv}: {dt:7.3f} ms/call, launches {kms:7.3f} ms" + (f" = {hbm / (kms * 1e-3) / 1e3:5.2f} TB/s of L" if trsv and kms > 0 else "")
    print(line, flush=True)
    h.close()

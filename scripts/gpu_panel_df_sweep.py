"""panel_df x dataflow tail width, mid sizes: a few fused look-ahead panels (their trailing updates on the 128-tile GEMM) in front of
a 64-tile dataflow tail, against the default schedule of each size.   python scripts/gpu_panel_df_sweep.py [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
sizes = [int(a) for a in sys.argv[1:]] or [6144, 8192, 10240, 12288, 14336, 16384]
def best_of(h, th, reps=5):
    h.loglik(th)
    b = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            h.loglik(th)
        b = min(b, (time.perf_counter() - t0) / reps)
    return b * 1e3
for n in sizes:
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    nt = (n + 127) // 128
    ref = h.loglik_parts(th)
    out = [f"default {best_of(h, th):6.2f}"]
    dmax = int(h.get_option("dataflow_max_nt"))
    for panel in (4,):
        for tail in (48, 64, 72, 80, 88, 96):
            if tail >= nt:
                continue
            h.set_option("panel_df", 1); h.set_option("panel", panel)
            h.set_option("dataflow_max_nt", min(dmax, nt - 1)); h.set_option("dataflow_tail", tail)
            r = h.loglik_parts(th)
            ok = r[3] == 0 and abs(r[0] - ref[0]) <= 1e-10 * abs(ref[0])
            out.append(f"P{panel}/t{tail} {best_of(h, th):6.2f}{'' if ok else ' WRONG'}")
    print(f"N={n:6d} | " + " | ".join(out), flush=True)
    h.close()

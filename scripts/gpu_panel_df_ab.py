"""Same-box A/B of the N = 11k-15k default (fused dataflow panels + 80-column tail) against panel_df = 0, and of the
one-workgroup-per-CU threshold of panel-restricted launches.   python scripts/gpu_panel_df_ab.py [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
sizes = [int(a) for a in sys.argv[1:]] or [11264, 12288, 13312, 14336, 15360]
def best_of(h, th, reps=5):
    h.loglik(th)
    b = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(reps):
            h.loglik(th)
        b = min(b, (time.perf_counter() - t0) / reps)
    return b * 1e3
for n in sizes:
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    out = []
    for name, opts in (("panel_df=0", {"panel_df": 0}), ("auto, one-wg<=600", {}), ("auto, one-wg<=2700", {"df_panel_one_wg_tasks": 2700}),
                       ("auto, one-wg<=1400", {"df_panel_one_wg_tasks": 1400}), ("auto, one-wg 0", {"df_panel_one_wg_tasks": 0})):
        for k, v in {"panel_df": -1, "df_panel_one_wg_tasks": 600, **opts}.items():
            h.set_option(k, v)
        out.append(f"{name}: {best_of(h, th):6.2f}")
    print(f"N={n:6d} | " + " | ".join(out), flush=True)
    h.close()

"""Option "panel_df" of the one-GPU look-ahead schedule (one theta, fp64): every outer panel -- look-ahead update + factorisation --
as ONE fused 64-tile dataflow launch on the panel stream, against the default (LA GEMM + three launches per tile column).
   python scripts/gpu_panel_df.py [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
sizes = [int(a) for a in sys.argv[1:]] or [10240, 12288, 14336, 16384, 20480, 24576, 32768]
for n in sizes:
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    h.set_option("dataflow_max_nt", min(int(h.get_option("dataflow_max_nt")), 64))      # (the look-ahead schedule also where one launch would do)
    out = []
    ref = None
    for name, opts in (("default", {}), ("panel_df", {"panel_df": 1}), ("panel_df,panel_wide=0", {"panel_df": 1, "panel_wide": 0}),
                       ("panel_df,tail48", {"panel_df": 1, "dataflow_tail": 48}), ("panel_df,tail96", {"panel_df": 1, "dataflow_tail": 96})):
        for k, v in {"panel_df": 0, "panel_wide": 1, "dataflow_tail": 64, **opts}.items():
            h.set_option(k, v)
        r = h.loglik_parts(th); h.loglik(th)
        reps = 5 if n <= 16384 else 3
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                h.loglik(th)
            best = min(best, (time.perf_counter() - t0) / reps)
        if ref is None:
            ref = r
        out.append(f"{name}: {best*1e3:7.2f} ms (dll {abs(r[0]-ref[0])/abs(ref[0]):.0e} info {r[3]})")
    print(f"N={n:6d} | " + " | ".join(out), flush=True)
    h.close()

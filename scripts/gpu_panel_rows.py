"""Batches: fused row-panel kernel (option panel_rows) vs the per-column in-panel update + solve launches: time per batch and
BIT-identity of every result.   python scripts/gpu_panel_rows.py [N B]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
cases = [(4096, 200, 64), (2048, 400, 64), (8192, 48, 64), (4096, 64, 64), (1024, 800, 64), (700, 300, 64), (4096, 100, 32), (3000, 64, 32)]
if len(sys.argv) > 2:
    cases = [(int(sys.argv[1]), int(sys.argv[2]), 64)]
for n, B, dtype in cases:
    d = 8
    kernel = "se_ard" if dtype == 64 else "matern52_ard"
    X, y = syn.make_dataset(n, d)
    Th = syn.theta_batch(B, kernel, d)
    Th[:, -1] = np.maximum(Th[:, -1], 0.05 if dtype == 64 else 0.3)
    Th[3, 0] = np.nan
    h = _lib.Handle(X, y, kernel, dtype=dtype)
    ref = None
    for rnd in range(2):
        row = [f"N={n} B={B} fp{dtype}"]
        for pr in (0, 1):
            for panel in (4, 2, 8):
                h.set_option("panel_rows", pr)
                h.set_option("panel", panel)
                h.loglik_batch(Th)
                t0 = time.perf_counter(); out, info = h.loglik_batch(Th); dt = time.perf_counter() - t0
                if pr == 0 and panel == 4 and ref is None:
                    ref = {}
                if pr == 0:
                    ref[panel] = (out.copy(), info.copy())
                ok = np.array_equal(info, ref[panel][1]) and np.array_equal(out[info == 0], ref[panel][0][ref[panel][1] == 0])
                row.append(f"rows={pr} panel={panel}: {dt*1e3:7.2f} ms {B*n**3/3/dt/1e12:5.1f} TF{'' if ok else ' MISMATCH'}")
        print(" | ".join(row), flush=True)
    if n == 4096 and B == 200:
        for pr in (0, 1):
            h.set_option("panel", 4); h.set_option("panel_rows", pr)
            h.set_option("profile", 2); h.reset_profile(); h.loglik_batch(Th)
            print(f"  profile panel_rows={pr}")
            for k, v in h.profile().items():
                if v["launches"]:
                    print(f"   {k:16s} {v['ms']:9.3f} ms  {int(v['launches']):5d} launches  {v['flops']/max(v['ms'],1e-9)/1e9:8.2f} TFLOP/s")
            h.set_option("profile", 0)
    h.close()

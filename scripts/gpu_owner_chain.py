"""The serial owner path of the sharded schedule with the chip to itself (potrf + panel solves + in-panel GEMMs of an evaluation
with look-ahead off and uniform 512-wide panels, as in scripts/gpu_multi_overhead.py) under variants of the panel kernels:
latency GEMM shape for large N (latency_max_nt), its tile threshold, fused potrf."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y = syn.make_dataset(n, 8)
th = syn.default_theta("se_ard", 8)
h = _lib.Handle(X, y, "se_ard")
h.set_option("lookahead", 0); h.set_option("panel_wide", 0); h.set_option("dataflow_tail", 0)
ref = None
for name, opts in (("base", {}), ("lat all N, <=256 tiles", {"latency_max_nt": 4096}), ("lat all N, <=512 tiles", {"latency_max_nt": 4096, "latency_tiles": 512}),
                   ("lat all N, <=1024 tiles", {"latency_max_nt": 4096, "latency_tiles": 1024}), ("left-looking panels", {"panel_left": 1}),
                   ("left + lat<=512", {"panel_left": 1, "latency_max_nt": 4096, "latency_tiles": 512}), ("panel 2", {"panel": 2}),
                   ("panel 2 + lat<=512", {"panel": 2, "latency_max_nt": 4096, "latency_tiles": 512})):
    for k, v in {"latency_max_nt": 48, "latency_tiles": 256, "panel_left": -1, "panel": 4}.items():
        h.set_option(k, v)
    for k, v in opts.items():
        h.set_option(k, v)
    h.set_option("profile", 0)
    ll, info = h.loglik(th)
    ref = ll if ref is None else ref
    h.set_option("profile", 2); h.reset_profile(); h.loglik(th)
    pr = h.profile()
    tot = pr["potrf"]["ms"] + pr["trsm"]["ms"] + pr["gemm_panel"]["ms"]
    print(f"{name:28s}: potrf {pr['potrf']['ms']:6.2f} ({int(pr['potrf']['launches'])}) + solves {pr['trsm']['ms']:6.2f} ({int(pr['trsm']['launches'])}) + in-panel {pr['gemm_panel']['ms']:6.2f} "
          f"({int(pr['gemm_panel']['launches'])}) = {tot:6.2f} ms; trailing {pr['syrk_trailing']['ms']:7.2f} ms; eval {pr['eval_total']['ms']:7.2f} ms; dll {ll - ref:.2e}", flush=True)
h.close()

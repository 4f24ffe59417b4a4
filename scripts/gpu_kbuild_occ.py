"""Kernel-build time (HIP events, profile class kbuild) of the CURRENT library at N = 32768, fp64 SE-ARD d = 8 and fp32 Matern
d = 16 -- one line per process, so that builds (GPHIP_LIB = a variant of scripts/ab_build.py) can be compared from a shell loop
(round 5: store order and occupancy of kbuild_mfma_kernel, profiles/r05_kbuild_store_order.txt, r05_kbuild_occupancy.txt):
   python scripts/gpu_kbuild_occ.py <tag>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
tag = sys.argv[1] if len(sys.argv) > 1 else "x"
out = []
for n, d, kernel, dtype in ((32768, 8, "se_ard", 64), (32768, 16, "matern52_ard", 32)):
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d, dtype="f32" if dtype == 32 else "f64")
    h = _lib.Handle(X, y, kernel, dtype=dtype)
    h.set_option("profile", 1)
    h.loglik(th)
    h.reset_profile()
    for i in range(6):
        h.loglik(th)
    p = h.profile()["kbuild"]
    ms = p["ms"] / max(int(p["launches"]), 1)
    gb = (8 if dtype == 64 else 4) * (n * (n + 1) / 2 + n * d) / 1e9
    out.append(f"fp{dtype} {ms:.3f} ms = {gb / ms / 8.0:.3f}")
    h.close()
print(f"{tag}: " + " | ".join(out), flush=True)

"""Kernel-matrix build (K1, BGP:29-43) of every named covariance family x input dimension x arithmetic type: HIP-event time of
the build launch inside likelihood evaluations at N = 32768 (profile class "kbuild") as a fraction of 8 TB/s -- algorithmic
bytes = s [N (N + 1) / 2 + N d] -- for the form the library's routing picks (option kbuild_mfma = 1) and for the direct form
(kbuild_mfma = 0).  Output -> profiles/r06_kbuild_family_table.txt
   python scripts/gpu_kbuild_family_table.py [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
fams = ["se", "se_ard", "matern52", "matern52_ard", "matern32", "matern32_ard", "rq", "rq_ard"]
print(f"# N = {n}; fraction of 8 TB/s: routed (kbuild_mfma = 1) / direct form (kbuild_mfma = 0)")
print("| family | type | " + " | ".join(f"d = {d}" for d in (1, 8, 16, 24, 32)) + " |")
print("|---|---|" + "---|" * 5)
for dtype in (64, 32):
    for fam in fams:
        cells = []
        for d in (1, 8, 16, 24, 32):
            X, y = syn.make_dataset(n, d)
            nl = d if fam.endswith("_ard") else 1
            ell = 0.3 if d == 1 else float(np.sqrt(d / 8.0))          # (r^2 stays O(1) as d grows)
            sn = 0.1 if dtype == 64 else 0.3
            th = [ell] * nl + ([2.0] if fam.startswith("rq") else []) + [1.0, sn]
            try:
                h = _lib.Handle(X, y, fam, dtype=dtype)
                h.set_option("profile", 1)
                res = {}
                for mode in (1, 0):
                    h.set_option("kbuild_mfma", mode)
                    ll, info = h.loglik(th)
                    assert info == 0, (fam, d, dtype, info)
                    h.reset_profile()
                    for _ in range(2):
                        h.loglik(th)
                    p = h.profile()["kbuild"]
                    res[mode] = p["ms"] / max(int(p["launches"]), 1)
                h.close()
                gb = (dtype // 8) * (n * (n + 1) / 2 + n * d) / 1e9
                cells.append(f"{gb / res[1] / 8.0:.2f} / {gb / res[0] / 8.0:.2f}")
            except Exception as e:
                cells.append("error: " + str(e)[:40])
        print(f"| {fam} | fp{dtype} | " + " | ".join(cells) + " |", flush=True)

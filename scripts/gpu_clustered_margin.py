"""Calibration of the conditioning-aware routing of the MFMA kernel build (gphip_ctx::kbuild_mfma_digits): on clustered
inputs (synthetic.make_clustered) the log-likelihood of the forced MFMA form (kbuild_mfma = 2) against the direct form
(0) and the oracle, next to the host's bound amp = eps max(B, 64) (1 + sf^2 / sn^2).  Output -> profiles/r06_clustered_margin.txt"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesianinference_amd import _lib, synthetic as syn
from oracle import gp_oracle as orc

EPS = 2.220446049250313e-16
print("# N d spread ell B sn noise | amp=eps*max(B,64)*(1+sf2/sn2) | rel|m2-m0| rel|m0-orc| rel|m2-orc| rel|m1-m0| (m1 took: direct/mfma) | ratio |m2-m0|/amp")
worst = 0.0
for n in (1500, 3000, 6000):
    for d in (1, 2, 3):
        for bt in (256.0, 500.0, 64.0, 8.0):
            for sn in (1e-3, 3e-3, 3e-2):
                for noise in (0.1, None):
                    for seed in (1, 2) if n <= 3000 else (1,):
                        nz = sn if noise is None else noise
                        X, y = syn.make_clustered(n, d, 100, 1e-4, nz, seed=syn.SEED + seed)
                        ell = np.sqrt(d / bt)
                        th = np.concatenate([np.full(d, ell), [1.0, sn]])
                        B = float(np.sum((np.ptp(X, axis=0) / 2 / ell) ** 2))
                        amp = EPS * max(B, 64.0) * (1 + 1 / sn ** 2)
                        want = orc.log_likelihood("se_ard", th, X, y)
                        h = _lib.Handle(X, y, "se_ard")
                        h.set_option("fused_eval", 0)
                        ll = {}
                        for mode in (0, 1, 2):
                            h.set_option("kbuild_mfma", mode)
                            ll[mode], info = h.loglik(th)
                            assert info == 0, (mode, info)
                        h.close()
                        r20 = abs(ll[2] - ll[0]) / abs(ll[0])
                        took = "direct" if ll[1] == ll[0] else "mfma"
                        worst = max(worst, r20 / amp)
                        print(f"{n} {d} 1e-4 {ell:.4f} {B:.0f} {sn:g} {nz:g} | {amp:.2e} | {r20:.2e} {abs(ll[0]-want)/abs(want):.2e} "
                              f"{abs(ll[2]-want)/abs(want):.2e} {abs(ll[1]-ll[0])/abs(ll[0]):.2e} {took} | {r20/amp:.3f}", flush=True)
print(f"# worst |m2-m0|/amp = {worst:.3f}")

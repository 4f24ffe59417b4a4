"""Generates tests/golden/*.npz from the CPU oracle (TEST INFRASTRUCTURE ONLY).

Run from the repo root:  python oracle/make_golden.py
The reference (Wolfram Language) cannot run here, so these vectors come from the oracle after it
has been pinned by closed forms, the MVN second formulation and mpmath (tests/test_oracle.py).
Fixtures: F1 cfg-1 (N=512,d=1,SE); F2 N=256,d=8 SE-ARD + Matern-5/2-ARD;
F3 scalars only for N=2048 ... 49152 (data regenerated from the seeded generator, checksummed; the N >= 16384 rows are
one in-place LU each: minutes of CPU, `--f3` regenerates only this file; `--f3b`: the N = 11k-15k companion);
F4 sentinel cases; HP mpmath 50-digit values for N=24/32/48 and 30-digit values for cfg 1 itself (N=512, d=1).
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bayesianinference_amd import synthetic as syn  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402
from oracle import hp_oracle as hp  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _evaluate(kernel, thetas, X, y, Xs, mean="zero", npred=4):
    ll, ld, qd, info = [], [], [], []
    for th in thetas:
        a, b, c, i = orc.log_likelihood(kernel, th, X, y, mean, parts=True)
        ll.append(a), ld.append(b), qd.append(c), info.append(i)
    mu, sd = [], []
    for th in thetas[:npred]:
        m, s = orc.predict_internal(kernel, th, X, y, Xs, mean)
        mu.append(m), sd.append(s)
    return dict(loglik=np.array(ll), logdet=np.array(ld), quad=np.array(qd), info=np.array(info),
                pred_mu=np.array(mu), pred_sd=np.array(sd))


def f1():
    X, y = syn.make_dataset(512, 1)
    Xs = np.linspace(-1.2, 1.2, 64)[:, None]
    thetas = syn.theta_batch(16, "se", 1)
    thetas[0] = syn.default_theta("se", 1)
    thetas[:, 2] = np.maximum(thetas[:, 2], 0.05)       # keep cond(K) <= ~1e8 for the parity bar
    res = _evaluate("se", thetas, X, y, Xs)
    np.savez_compressed(os.path.join(OUT, "f1_se_n512_d1.npz"), X=X, y=y, Xs=Xs, thetas=thetas,
                        kernel="se", mean="zero", **res)


def f2():
    X, y = syn.make_dataset(256, 8)
    Xs = syn.make_test_points(64, 8)
    for kernel in ("se_ard", "matern52_ard"):
        thetas = syn.theta_batch(16, kernel, 8)
        thetas[0] = syn.default_theta(kernel, 8)
        thetas[:, -1] = np.maximum(thetas[:, -1], 0.05)
        res = _evaluate(kernel, thetas, X, y, Xs)
        np.savez_compressed(os.path.join(OUT, f"f2_{kernel}_n256_d8.npz"), X=X, y=y, Xs=Xs,
                            thetas=thetas, kernel=kernel, mean="zero", **res)
    # constant-mean variant + isotropic matern, ragged N (not a multiple of the 128 tile)
    X, y = syn.make_dataset(333, 3)
    Xs = syn.make_test_points(17, 3)
    thetas = np.column_stack([syn.theta_batch(8, "matern52", 3), np.linspace(-0.5, 0.5, 8)])
    thetas[:, 2] = np.maximum(thetas[:, 2], 0.05)
    res = _evaluate("matern52", thetas, X, y, Xs, mean="const")
    np.savez_compressed(os.path.join(OUT, "f2_matern52_const_n333_d3.npz"), X=X, y=y, Xs=Xs,
                        thetas=thetas, kernel="matern52", mean="const", **res)


def _f3_rows(sizes, fname):
    rows = []
    for n, kernel in sizes:
        X, y = syn.make_dataset(n, 8)
        th = syn.default_theta(kernel, 8)
        ll, ld, qd, info = orc.log_likelihood(kernel, th, X, y, parts=True)
        rows.append((n, 8, kernel, float(X.sum()), float(y.sum()), ll, ld, qd, info))
        print("F3", rows[-1])
    np.savez_compressed(
        os.path.join(OUT, fname),
        n=np.array([r[0] for r in rows]), d=np.array([r[1] for r in rows]),
        kernel=np.array([r[2] for r in rows]),
        xsum=np.array([r[3] for r in rows]), ysum=np.array([r[4] for r in rows]),
        loglik=np.array([r[5] for r in rows]), logdet=np.array([r[6] for r in rows]),
        quad=np.array([r[7] for r in rows]), info=np.array([r[8] for r in rows]))


def f3():
    # N = 16384 / 32768 / 49152: the sizes at which the look-ahead / wide-panel / dataflow-tail schedule of the HIP path
    # is active (N = 32768 is the BASELINE metric's own size): LU in place, 8.6 GB / 19.3 GB of matrix, minutes of CPU
    sizes = [(2048, "se_ard"), (4096, "se_ard"), (8192, "se_ard"), (2048, "matern52_ard"), (16384, "se_ard"),
             (32768, "se_ard"), (16384, "matern52_ard")]
    if os.environ.get("GOLDEN_F3_HUGE") == "1":
        sizes.append((49152, "se_ard"))
    _f3_rows(sizes, "f3_scalars.npz")


def f3b():
    # N = 11k-15k: the range in which the HIP path runs fused dataflow panels in front of an 80-column dataflow tail
    # (option panel_df, by size; round 4) -- its own file so that the minutes-long f3 need not be regenerated
    _f3_rows([(12288, "se_ard"), (13440, "matern52_ard"), (11300, "se_ard")], "f3b_scalars.npz")


def fcfg5():
    """BASELINE.json cfg 5 at its own size: Matern-5/2 ARD, N = 65536, d = 16, the fp32 timing theta (sn = 0.3) -- ONE in-place
    LU of a 34.4 GB matrix (needs ~36 GB of RAM and ~15-40 min on 8 cores; `--cfg5` regenerates only this file).  The three
    scalars AND, from the same factorisation, the prediction (BGP:396-422: mu*, sigma*) at the first 64 of cfg 5's 10 000 test
    points -- the steps of gp_oracle.predict_internal with the LU shared instead of redone."""
    n, d, kernel, m = 65536, 16, "matern52_ard", 64
    X, y = syn.make_dataset(n, d)
    Xs = syn.make_test_points(10000, d)[:m]
    th = syn.default_theta(kernel, d, dtype="f32")
    r = orc.residual(kernel, th, X, y)
    K = orc.covariance_matrix(kernel, th, X)
    solve, ld = orc.matrix_inverse_and_det(K.T, overwrite=True)      # (symmetric bit for bit: the transpose view is column-major)
    del K
    alpha = solve(r)
    qd = float(r @ alpha)
    ll = orc.gp_log_likelihood_from_parts(r, solve, ld)
    k, kappa = orc.k_and_kappa(kernel, th, X, Xs)
    mu = alpha @ k                                                    # mean function 0 (BGP:407-412)
    with np.errstate(invalid="ignore"):
        sd = np.sqrt(kappa - np.sum(k * solve(k), axis=0))            # BGP:414-417
    print("CFG5", n, d, kernel, ll, ld, qd, mu[:3], sd[:3])
    np.savez_compressed(os.path.join(OUT, "cfg5_scalars.npz"), n=n, d=d, kernel=kernel, theta=th, xsum=float(X.sum()),
                        ysum=float(y.sum()), loglik=ll, logdet=ld, quad=qd, info=0, Xs=Xs, mu=mu, sd=sd)


def f4():
    # duplicate rows + zero nugget -> singular; sigma_n = 1e-12 -> hopelessly ill-conditioned
    X, y = syn.make_dataset(64, 2)
    Xd = X.copy()
    Xd[17] = Xd[3]
    cases = dict(
        dup_X=Xd, dup_y=y, dup_theta=np.array([1.0, 1.0, 1.0, 0.0]),
        ill_X=X, ill_y=y, ill_theta=np.array([5.0, 5.0, 1.0, 1e-12]),
        ok_X=X, ok_y=y, ok_theta=np.array([0.5, 0.7, 1.3, 0.2]))
    out = {}
    for name in ("dup", "ill", "ok"):
        ll, ld, qd, info = orc.log_likelihood("se_ard", cases[f"{name}_theta"], cases[f"{name}_X"],
                                              cases[f"{name}_y"], parts=True)
        out[f"{name}_loglik"], out[f"{name}_info"] = ll, info
        print("F4", name, ll, info)
    np.savez_compressed(os.path.join(OUT, "f4_sentinel.npz"), kernel="se_ard", **cases, **out)


def fhp():
    out = {}
    for n, d, kernel in ((24, 2, "se_ard"), (48, 3, "matern52_ard"), (32, 1, "se")):
        X, y = syn.make_dataset(n, d)
        Xs = syn.make_test_points(5, d)
        th = syn.default_theta(kernel, d)
        th[-1] = 0.2
        ll, ld, qd = hp.log_likelihood(kernel, th, X.tolist(), y.tolist())
        mu, sd = hp.predict(kernel, th, X.tolist(), y.tolist(), Xs.tolist())
        key = f"{kernel}_n{n}"
        out.update({f"{key}_X": X, f"{key}_y": y, f"{key}_Xs": Xs, f"{key}_theta": th,
                    f"{key}_loglik": ll, f"{key}_logdet": ld, f"{key}_quad": qd,
                    f"{key}_mu": np.array(mu), f"{key}_sd": np.array(sd)})
        print("HP", key, ll)
    # cfg 1's own size (N = 512, d = 1, BASELINE.json configs[0]) at 30 digits: plain-list Cholesky, ~20 s
    X, y = syn.make_dataset(512, 1)
    Xs = np.linspace(-1.2, 1.2, 7)[:, None]
    th = syn.default_theta("se", 1)
    ll, ld, qd, mu, sd = hp.cholesky_pin("se", th, X.tolist(), y.tolist(), Xs.tolist(), dps=30)
    key = "se_n512"
    out.update({f"{key}_X": X, f"{key}_y": y, f"{key}_Xs": Xs, f"{key}_theta": th, f"{key}_loglik": ll, f"{key}_logdet": ld,
                f"{key}_quad": qd, f"{key}_mu": np.array(mu), f"{key}_sd": np.array(sd)})
    print("HP", key, ll)
    np.savez_compressed(os.path.join(OUT, "hp_mpmath.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--cfg5" in sys.argv:
        fcfg5()
        sys.exit(0)
    if "--f3" in sys.argv:
        f3()
        sys.exit(0)
    if "--f3b" in sys.argv:
        f3b()
        sys.exit(0)
    f1(), f2(), f4(), fhp(), f3(), f3b()
    print("golden fixtures written to", OUT)

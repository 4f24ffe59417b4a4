"""Kernel build with the distance cross term on the matrix pipe (kbuild_mfma_kernel, option kbuild_mfma) against the direct
form (kbuild_kernel), one process, one box:
  * HIP-event time of the build launches inside likelihood evaluations at N = 32768 (profile class "kbuild"), both forms,
    interleaved, as a fraction of 8 TB/s;
  * K and k* at N = 700 against numpy on the kernel's definition and against each other (largest relative difference), including length scales
    at which the host's bound sends the slot back to the direct kernel;
  * a batch whose thetas straddle the bound (mixed launch) against the all-direct batch.
   python scripts/gpu_kbuild_mfma.py [quick]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn


def kmat(kernel, th, X):
    """K(theta) in numpy from the kernel's definition -- a developer script does not touch oracle/"""
    d = X.shape[1]
    nl = d if kernel.endswith("_ard") else 1
    ell, sf, sn = np.broadcast_to(th[:nl], (d,)), th[nl], th[nl + 1]
    U = X / ell
    r2 = np.maximum(((U[:, None, :] - U[None, :, :]) ** 2).sum(-1), 0.0)
    if kernel.startswith("se"):
        K = sf * sf * np.exp(-0.5 * r2)
    else:
        s5 = np.sqrt(5.0 * r2)
        K = sf * sf * (1.0 + s5 + 5.0 * r2 / 3.0) * np.exp(-s5)
    return K + sn * sn * np.eye(len(X))

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def timing():
    cases = ((32768, 8, "se_ard", 64, 6), (32768, 16, "matern52_ard", 32, 4), (32768, 8, "matern52_ard", 64, 3),
             (32768, 16, "se_ard", 32, 3), (32768, 1, "se", 64, 3), (32768, 3, "se_ard", 64, 3), (32768, 24, "se_ard", 64, 3))
    if quick:
        cases = cases[:2]
    for n, d, kernel, dtype, reps in cases:
        X, y = syn.make_dataset(n, d)
        th = syn.default_theta(kernel, d, dtype="f32" if dtype == 32 else "f64")
        h = _lib.Handle(X, y, kernel, dtype=dtype)
        h.set_option("profile", 1)
        res = {}
        lls = {}
        for rnd in range(2):
            for mode in (0, 1):
                h.set_option("kbuild_mfma", mode)
                h.loglik(th)
                h.reset_profile()
                for i in range(reps):
                    lls[mode] = h.loglik(th)[0]
                p = h.profile()["kbuild"]
                res[mode] = p["ms"] / max(int(p["launches"]), 1)
        es = 8 if dtype == 64 else 4
        gb = es * (n * (n + 1) / 2 + n * d) / 1e9
        print(f"kbuild {kernel} fp{dtype} N={n} d={d}: direct {res[0]:.3f} ms = {gb / res[0] / 8.0:.3f} of 8 TB/s | mfma {res[1]:.3f} ms = "
              f"{gb / res[1] / 8.0:.3f} | loglik rel diff {abs(lls[0] - lls[1]) / abs(lls[0]):.2e}", flush=True)
        h.close()


def accuracy():
    for kernel, dtype, d, scale in (("se_ard", 64, 8, 1.0), ("se_ard", 64, 8, 0.2), ("se_ard", 64, 8, 0.05), ("matern52_ard", 64, 8, 1.0),
                                    ("matern52_ard", 64, 8, 0.2), ("se", 64, 1, 1.0), ("se", 64, 1, 0.05), ("se_ard", 64, 5, 1.0),
                                    ("se_ard", 64, 13, 1.0), ("se_ard", 64, 24, 2.0), ("matern52_ard", 32, 16, 1.0), ("se_ard", 32, 16, 1.0),
                                    ("se_ard", 32, 3, 0.3)):
        X, y = syn.make_dataset(700, d)
        X = X + 3.0                                    # (off-centre inputs: the kernel's centring has to earn its keep)
        th = syn.default_theta(kernel, d, dtype="f32" if dtype == 32 else "f64")
        nl = d if kernel.endswith("_ard") else 1
        th[:nl] *= scale
        Xs = syn.make_test_points(300, d) + 3.0
        Ko = kmat(kernel, th, X)
        h = _lib.Handle(X, y, kernel, dtype=dtype)
        out = {}
        for mode in (0, 1, 2):
            h.set_option("kbuild_mfma", mode)
            K = h.covariance(th)
            k, kappa = h.cross_covariance(th, Xs)
            out[mode] = (K, k)
        h.close()
        ref = np.abs(Ko).max()
        e = {m: np.abs(out[m][0] - Ko).max() / ref for m in out}
        rel = {m: (np.abs(out[m][0] - Ko) / np.abs(Ko).clip(1e-300)).max() for m in out}
        dk = {m: np.abs(out[m][1] - out[0][1]).max() / ref for m in out}
        same = np.array_equal(out[0][0], out[1][0])
        print(f"K {kernel} fp{dtype} d={d} l*{scale}: |K - numpy| / max: direct {e[0]:.2e} auto {e[1]:.2e} forced {e[2]:.2e}; entrywise rel "
              f"{rel[0]:.2e} / {rel[1]:.2e} / {rel[2]:.2e}; auto == direct bitwise: {same}; cross vs direct: auto {dk[1]:.2e} forced {dk[2]:.2e}",
              flush=True)


def batch():
    n, d = 1500, 8
    X, y = syn.make_dataset(n, d)
    rng = np.random.default_rng(3)
    B = 24
    Th = np.tile(syn.default_theta("se_ard", d), (B, 1))
    Th[:, :d] *= np.exp(rng.uniform(np.log(0.03), np.log(3.0), size=(B, d)))
    h = _lib.Handle(X, y, "se_ard")
    r = {}
    for mode in (0, 1):
        h.set_option("kbuild_mfma", mode)
        r[mode] = h.loglik_batch(Th)
    h.close()
    ll0, i0 = r[0]
    ll1, i1 = r[1]
    bound = ((1.0 / Th[:, :d]) ** 2).sum(axis=1)
    print("batch: info equal:", np.array_equal(i0, i1), " max rel loglik diff:", np.max(np.abs(ll0 - ll1) / np.abs(ll0)),
          " slots under the bound:", int((bound <= 512).sum()), "of", B, flush=True)


accuracy()
batch()
timing()

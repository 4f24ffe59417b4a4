"""CPU ORACLE (high precision) -- TEST INFRASTRUCTURE ONLY.

50-digit mpmath evaluation of the GP log marginal likelihood and prediction formulas of
BayesianGaussianProcess.wl:181-199 and :396-422, used to pin `oracle/gp_oracle.py` for
N <= 64 (SURVEY.md §8c "High-precision pin").  Pure-Python loops: small cases only.
"""
from __future__ import annotations

import mpmath as mp

mp.mp.dps = 50


def _kernel(kernel, ell, sf, a, b):
    r2 = mp.mpf(0)
    for j in range(len(a)):
        t = (mp.mpf(float(a[j])) - mp.mpf(float(b[j]))) / ell[j]
        r2 += t * t
    if kernel in ("se", "se_ard"):
        return sf * sf * mp.exp(-r2 / 2)
    if kernel in ("matern52", "matern52_ard"):
        s5 = mp.sqrt(5) * mp.sqrt(r2)
        return sf * sf * (1 + s5 + mp.mpf(5) / 3 * r2) * mp.exp(-s5)
    raise ValueError(kernel)


def general_log_likelihood(family, ell, sf, sn, X, y, alpha=None, offset=0.0):
    """The same 50-digit likelihood for ONE term of the general kernel grammar of oracle/gp_oracle.py (round 6: the families that
    got a matrix-pipe build on the device) -- `family` in se / matern52 / matern32 / rq, `ell` one length scale per dimension,
    `alpha` the rational quadratic's shape, `offset` the additive constant c of "term + const" (BGP:16 is c + SE).  Pins
    gp_oracle.general_kernel_matrix / log_likelihood for those forms."""
    n, d = len(X), len(X[0])
    ell = [mp.mpf(float(v)) for v in ell]
    sf, sn, c = mp.mpf(float(sf)), mp.mpf(float(sn)), mp.mpf(float(offset))

    def k(a, b):
        r2 = mp.mpf(0)
        for j in range(d):
            t = (mp.mpf(float(a[j])) - mp.mpf(float(b[j]))) / ell[j]
            r2 += t * t
        if family == "se":
            g = mp.exp(-r2 / 2)
        elif family == "matern52":
            s5 = mp.sqrt(5 * r2)
            g = (1 + s5 + mp.mpf(5) / 3 * r2) * mp.exp(-s5)
        elif family == "matern32":
            s3 = mp.sqrt(3 * r2)
            g = (1 + s3) * mp.exp(-s3)
        elif family == "rq":
            a_ = mp.mpf(float(alpha))
            g = (1 + r2 / (2 * a_)) ** (-a_)
        else:
            raise ValueError(family)
        return sf * sf * g + c

    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(i + 1):
            K[i, j] = K[j, i] = k(X[i], X[j])
        K[i, i] += sn * sn
    L = mp.cholesky(K)
    z = mp.lu_solve(L, mp.matrix([mp.mpf(float(v)) for v in y]))
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(n))
    quad = sum(z[i] * z[i] for i in range(n))
    return float(-(n * mp.log(2 * mp.pi) + logdet + quad) / 2), float(logdet), float(quad)


def _split(kernel, d, theta, mean):
    th = [mp.mpf(float(t)) for t in theta]
    nl = 1 if kernel in ("se", "matern52") else d
    ell = [th[0]] * d if nl == 1 else th[:nl]
    sf, sn = th[nl], th[nl + 1]
    mu = th[nl + 2] if mean == "const" else mp.mpf(0)
    return ell, sf, sn, mu


def _cov(kernel, theta, X, mean):
    n, d = len(X), len(X[0])
    ell, sf, sn, mu = _split(kernel, d, theta, mean)
    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(i + 1):
            K[i, j] = K[j, i] = _kernel(kernel, ell, sf, X[i], X[j])
        K[i, i] += sn * sn
    return K, (ell, sf, sn, mu)


def log_likelihood(kernel, theta, X, y, mean="zero"):
    """-(N log 2pi + log det K + r.K^-1 r)/2 (BGP:190-196) at 50 digits via Cholesky."""
    K, (_, _, _, mu) = _cov(kernel, theta, X, mean)
    n = len(X)
    L = mp.cholesky(K)
    r = mp.matrix([mp.mpf(float(v)) - mu for v in y])
    z = mp.lu_solve(L, r)          # L is triangular; lu_solve is exact enough at 50 dps
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(n))
    quad = sum(z[i] * z[i] for i in range(n))
    ll = -(n * mp.log(2 * mp.pi) + logdet + quad) / 2
    return float(ll), float(logdet), float(quad)


def predict(kernel, theta, X, y, Xs, mean="zero"):
    """mu*, sigma* of BGP:407-417 at 50 digits."""
    K, (ell, sf, sn, mu) = _cov(kernel, theta, X, mean)
    n = len(X)
    r = mp.matrix([mp.mpf(float(v)) - mu for v in y])
    alpha = mp.lu_solve(K, r)
    mus, sds = [], []
    for xs in Xs:
        k = mp.matrix([_kernel(kernel, ell, sf, X[i], xs) for i in range(n)])
        v = mp.lu_solve(K, k)
        kappa = sf * sf + sn * sn
        mus.append(float(mu + sum(alpha[i] * k[i] for i in range(n))))
        sds.append(float(mp.sqrt(kappa - sum(k[i] * v[i] for i in range(n)))))
    return mus, sds


def cholesky_pin(kernel, theta, X, y, Xs=(), mean="zero", dps=30):
    """The same formulas for a problem of a few hundred points (cfg 1: N = 512, d = 1) -- arithmetic-independent
    truth for the size the reference's own CPU-runnable configuration has.  Plain-list Cholesky with mp.fdot inner
    products (mp.matrix element access would dominate), forward substitution for z = L^-1 r and the test points.
    Returns (loglik, logdet, quad, mu*[..], sd*[..]) as floats; `dps` significant digits (30 leaves > 10 digits of
    margin over fp64 at cond(K) ~ 1e5)."""
    old = mp.mp.dps
    mp.mp.dps = dps
    try:
        n, d = len(X), len(X[0])
        ell, sf, sn, mu = _split(kernel, d, theta, mean)
        L = [[None] * (i + 1) for i in range(n)]
        for i in range(n):
            for j in range(i + 1):
                L[i][j] = _kernel(kernel, ell, sf, X[i], X[j])
            L[i][i] += sn * sn
        for j in range(n):
            Lj = L[j]
            dj = mp.sqrt(Lj[j] - mp.fdot(Lj[:j], Lj[:j]))
            Lj[j] = dj
            for i in range(j + 1, n):
                Li = L[i]
                Li[j] = (Li[j] - mp.fdot(Li[:j], Lj[:j])) / dj
        r = [mp.mpf(float(v)) - mu for v in y]

        def forward(b):
            z = [None] * n
            for i in range(n):
                z[i] = (b[i] - mp.fdot(L[i][:i], z[:i])) / L[i][i]
            return z

        z = forward(r)
        logdet = 2 * sum(mp.log(L[i][i]) for i in range(n))
        quad = mp.fdot(z, z)
        ll = -(n * mp.log(2 * mp.pi) + logdet + quad) / 2
        mus, sds = [], []
        for xs in Xs:
            k = [_kernel(kernel, ell, sf, X[i], xs) for i in range(n)]
            v = forward(k)
            mus.append(float(mu + mp.fdot(v, z)))
            sds.append(float(mp.sqrt(sf * sf + sn * sn - mp.fdot(v, v))))
        return float(ll), float(logdet), float(quad), mus, sds
    finally:
        mp.mp.dps = old

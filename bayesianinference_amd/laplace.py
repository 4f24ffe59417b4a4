"""Laplace approximation of the evidence for a GP inferenceObject -- the caller the hyper-parameter gradient was built
for.  Host-side restatement of `approximateEvidence` / `laplaceLogEvidence`
(LaplaceApproximation.wl:22-30, 177-238, cited LA:line):

    maximise  log posterior = log-likelihood + log prior  over the parameter box      LA:192-210 (NMaximize / FindMaximum)
    precision = - Hessian of the log posterior at the maximum                           LA:215-217 (numericD .. "Hessian")
    log Z    ~= max + (p log 2 pi - log det precision) / 2                              LA:23-28

MI355X-first difference: the reference maximises without derivatives and differentiates the density numerically twice;
here the likelihood AND its gradient come from ONE device call (gphip_loglik_grad: one factorisation, K^-1 on the MFMA),
the maximiser is L-BFGS-B on the box, and the Hessian is a central difference of GRADIENTS (2p gradient calls instead
of O(p^2) density calls).  All dense arithmetic stays in the HIP library; nothing here falls back to the CPU.
"""
from __future__ import annotations

import math

import numpy as np

from .gaussian_process import MACHINE_LOG_ZERO, inferenceObject


def laplaceLogEvidence(maximum: float, precision) -> float | None:
    """LA:22-30: max + (p log 2pi - log det P)/2; None (Missing[]) unless det P > 0."""
    P = np.atleast_2d(np.asarray(precision, dtype=np.float64))
    sign, logdet = np.linalg.slogdet(P)
    if not (sign > 0 and math.isfinite(logdet)):
        return None
    return float(maximum + 0.5 * (P.shape[0] * math.log(2.0 * math.pi) - logdet))


def _prior_grad(logprior, theta, lo, hi, rel=1e-6):
    """central differences of the (cheap, host-side) log prior; one-sided at the box faces"""
    g = np.zeros_like(theta)
    for k in range(len(theta)):
        step = rel * max(abs(theta[k]), hi[k] - lo[k])
        a, b = theta.copy(), theta.copy()
        a[k], b[k] = min(theta[k] + step, hi[k]), max(theta[k] - step, lo[k])
        fa, fb = logprior(a), logprior(b)
        g[k] = (fa - fb) / (a[k] - b[k]) if a[k] > b[k] and min(fa, fb) > MACHINE_LOG_ZERO else 0.0
    return g


def approximateEvidence(obj, InitialGuess=None, Starts: int = 4, HessianStep: float = 1e-4, Seed: int = 0):
    """LA:177-238 for an object that carries "LogLikelihoodGradientFunction" (every HIP-backed GP object does).
    InitialGuess: a theta to start from (the reference's FindMaximum branch, LA:193-203); otherwise `Starts` random
    starts in the box stand in for NMaximize's global search (LA:204-209).  Returns the reference's association:
    "LogEvidence" (absent if the precision matrix is not positive definite, LA:218-226), "Maximum" = (value, theta),
    "Mean", "PrecisionMatrix", "Parameters"; None ($Failed) if no start converges to a finite maximum."""
    from scipy.optimize import minimize
    if not isinstance(obj, inferenceObject) or obj.failed or "LogLikelihoodGradientFunction" not in obj:
        return None
    params = obj["Parameters"]
    lo = np.array([p[1] for p in params], dtype=np.float64)
    hi = np.array([p[2] for p in params], dtype=np.float64)
    value_grad, logprior = obj["LogLikelihoodGradientFunction"], obj["LogPriorPDFFunction"]

    def neg_post(theta):
        theta = np.clip(theta, lo, hi)
        ll, g = value_grad(theta)
        lp = logprior(theta)
        if ll <= MACHINE_LOG_ZERO or lp <= MACHINE_LOG_ZERO or not np.all(np.isfinite(g)):
            return 1e300, np.zeros_like(theta)               # the sentinel: a wall, never an exception
        return -(ll + lp), -(g + _prior_grad(logprior, theta, lo, hi))

    rng = np.random.default_rng(Seed)
    starts = [np.asarray(InitialGuess, dtype=np.float64)] if InitialGuess is not None else \
        [np.exp(np.log(lo) + rng.random(len(lo)) * (np.log(hi) - np.log(lo))) if np.all(lo > 0)
         else lo + rng.random(len(lo)) * (hi - lo) for _ in range(max(1, Starts))]
    best = None
    for x0 in starts:
        res = minimize(neg_post, np.clip(x0, lo, hi), jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)))
        if math.isfinite(res.fun) and res.fun < 1e299 and (best is None or res.fun < best.fun):
            best = res
    if best is None:
        return None                                          # approximateEvidence::nmaximize
    mean, maximum = np.clip(best.x, lo, hi), -float(best.fun)
    p = len(mean)
    H = np.zeros((p, p))
    for k in range(p):                                       # Hessian = central difference of gradients
        step = HessianStep * max(abs(mean[k]), 1e-3 * (hi[k] - lo[k]))
        a, b = mean.copy(), mean.copy()
        a[k], b[k] = min(mean[k] + step, hi[k]), max(mean[k] - step, lo[k])
        H[:, k] = (neg_post(a)[1] - neg_post(b)[1]) / (a[k] - b[k])
    precision = 0.5 * (H + H.T)                              # - Hessian of the log posterior (neg_post is already negated)
    out = {"Maximum": (maximum, mean), "Mean": mean, "PrecisionMatrix": precision,
           "Parameters": [p_[0] for p_ in params]}
    logz = laplaceLogEvidence(maximum, precision)
    if logz is not None:
        out["LogEvidence"] = logz
    return out

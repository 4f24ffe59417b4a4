"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A numpy/scipy fp64 restatement of the Gaussian-process path of ssmit1986/BayesianInference
(`BayesianInference/Kernel/BayesianGaussianProcess.wl`, cited below as BGP:line, and
`BayesianUtilities.wl` as BU:line).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module; the product (`bayesianinference_amd`)
never does and fails loudly when its HIP library is missing.

PARITY PIN STATUS: the reference is Wolfram Language and cannot execute here or on the GPU box
(no Wolfram kernel), and it ships no tests, golden vectors or fixtures.  The
arithmetic it delegates to (``LinearSolve`` = LAPACK LU, ``Compile``) is closed source.  This
oracle is therefore pinned by:
  * closed-form known answers derived from BGP:181-199 (N=1, N=2, null kernel, far/near limits),
  * the second formulation the reference itself sanctions -- its ``Automatic`` branch is the
    multivariate-normal log-pdf (BGP:273-292) -> scipy.stats.multivariate_normal.logpdf,
  * 50-digit mpmath evaluation (`oracle/hp_oracle.py`) for N <= 64,
all checked in `tests/test_oracle.py` and frozen into `tests/golden/*.npz` by
`oracle/make_golden.py`.  No output of the real reference exists to compare with: by the task's
definition this is "parity unpinned by the reference's own vectors" and DESIGN.md says so.

Like the reference, this oracle factors K with partially pivoted **LU** (``LinearSolve``,
BGP:130-136), not Cholesky; the HIP product path uses Cholesky.  Agreement of the two is
part of what the parity tests establish.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla

# BU:47  $MachineLogZero = -Statistics`Library`MachineInfinity.  The constant is closed source; the
# WL shim reads it at load time.  Here: the most negative finite double (any value that is finite,
# hugely negative and survives Clip works identically for the sampler, BS:276-298).
MACHINE_LOG_ZERO = -1.7976931348623157e308
LOG_TWO_PI = math.log(2.0 * math.pi)          # BGP:182

KERNELS = ("se", "se_ard", "matern52", "matern52_ard", "null")


class MatInvFailure(Exception):
    """Stands for Throw[$MachineLogZero, "MatInv"] (BGP:131-135)."""


# --------------------------------------------------------------------------------------
# Named kernels.  theta layout: (l_1..l_nl, sigma_f, sigma_n [, mu])
# --------------------------------------------------------------------------------------
def n_lengthscales(kernel: str, d: int) -> int:
    if kernel in ("se", "matern52"):
        return 1
    if kernel in ("se_ard", "matern52_ard"):
        return d
    if kernel == "null":
        return 0
    raise ValueError(kernel)


def n_params(kernel: str, d: int, mean: str = "zero") -> int:
    # null kernel: theta = (sigma_n [, mu])
    base = 1 if kernel == "null" else n_lengthscales(kernel, d) + 2
    return base + (1 if mean == "const" else 0)


def split_theta(kernel: str, d: int, theta, mean: str = "zero"):
    theta = np.asarray(theta, dtype=np.float64)
    nl = n_lengthscales(kernel, d)
    if kernel == "null":
        ell, sf, sn = np.ones(1), 0.0, theta[0]
        mu = theta[1] if mean == "const" else 0.0
        return ell, sf, sn, mu
    ell = np.broadcast_to(theta[:nl], (d,)) if nl == 1 else theta[:nl]
    sf, sn = theta[nl], theta[nl + 1]
    mu = theta[nl + 2] if mean == "const" else 0.0
    return np.asarray(ell, dtype=np.float64), float(sf), float(sn), float(mu)


def kernel_matrix(kernel: str, ell, sf: float, A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """k(a_i, b_j) for all pairs: the `kernel @@ points[[{i,j}]]` evaluations of BGP:32 /
    `Table[kernel[i,j], {i,points1},{j,points2}]` of BGP:100-109.  Direct sum of squared scaled
    differences (no |a|^2+|b|^2-2ab expansion: cancellation would cost the 1e-8 parity)."""
    A = np.asarray(A, dtype=np.float64) / ell
    B = np.asarray(B, dtype=np.float64) / ell
    if A.shape[0] * B.shape[0] > _BLOCK_ELEMS:
        # large problems (golden scalars at N = 16384 / 32768, the bench's CPU baseline): the same arithmetic,
        # element for element, in row blocks so that the temporaries stay small next to the result
        out = np.empty((A.shape[0], B.shape[0]))
        step = max(1, _BLOCK_ELEMS // B.shape[0])
        for i0 in range(0, A.shape[0], step):
            out[i0:i0 + step] = _kernel_block(kernel, sf, A[i0:i0 + step], B)
        return out
    return _kernel_block(kernel, sf, A, B)


_BLOCK_ELEMS = 1 << 26          # 512 MiB of fp64 per temporary


def _kernel_block(kernel: str, sf: float, A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """A, B already divided by the length scales."""
    r2 = np.zeros((A.shape[0], B.shape[0]))
    for j in range(A.shape[1]):          # O(N*M) memory, not O(N*M*d)
        diff = A[:, j][:, None] - B[:, j][None, :]
        r2 += diff * diff
    if kernel in ("se", "se_ard"):
        return (sf * sf) * np.exp(-0.5 * r2)
    if kernel in ("matern52", "matern52_ard"):
        s = np.sqrt(r2)
        s5 = math.sqrt(5.0) * s
        return (sf * sf) * (1.0 + s5 + (5.0 / 3.0) * r2) * np.exp(-s5)
    if kernel == "null":
        return np.zeros_like(r2)
    raise ValueError(kernel)


def covariance_matrix(kernel: str, theta, X: np.ndarray, mean: str = "zero", nugget_fn=None):
    """BGP:27-43 covarianceMatrix: K_ij = k(x_i,x_j) + delta_ij nugget(x_i).
    Null kernel (BGP:25-27): returns the diagonal *vector* nugget /@ points.
    nugget_fn: optional function of ONE point (`nugget[points[[i]]]`, BGP:37) returning a variance; None = the constant
    Function[sn^2]."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    ell, sf, sn, _ = split_theta(kernel, X.shape[1], theta, mean)
    nug = np.full(X.shape[0], sn * sn) if nugget_fn is None else np.array([float(nugget_fn(x)) for x in X])
    if kernel == "null":
        return nug
    K = kernel_matrix(kernel, ell, sf, X, X)
    K[np.diag_indices_from(K)] += nug
    return K


# --------------------------------------------------------------------------------------
# BGP:126-159 matrixInverseAndDet
# --------------------------------------------------------------------------------------
RCOND_FAIL = 2.220446049250313e-16   # stands for the LinearSolve::luc warning threshold


def matrix_inverse_and_det(matrix: np.ndarray, overwrite: bool = False):
    """Returns (solve, logdet).  Dense: LU (``LinearSolve[matrix]``, BGP:132), log-det =
    Total@Log@Abs@Diagonal[U] (BGP:126-128,139).  Vector argument = diagonal matrix (BGP:156-159).
    Raises MatInvFailure where the reference Throws on LinearSolve::sing1 / ::luc."""
    matrix = np.asarray(matrix, dtype=np.float64)
    if matrix.ndim == 1:                                   # BGP:156-159
        diag = matrix
        return (lambda b: (b.T / diag).T if b.ndim == 2 else b / diag), float(np.sum(np.log(np.abs(diag))))
    if not np.all(np.isfinite(matrix)):
        raise MatInvFailure("non-finite covariance")
    anorm = float(np.abs(matrix).sum(axis=0).max()) if not overwrite else _norm1_blocked(matrix)
    # overwrite: factor in place (the caller gives the matrix up) -- large-N golden scalars / CPU baseline only
    lu, piv = sla.lu_factor(matrix, overwrite_a=overwrite, check_finite=False)
    u = np.diag(lu)
    if np.any(u == 0.0) or not np.all(np.isfinite(u)):
        raise MatInvFailure("sing1")
    rcond, _ = sla.lapack.dgecon(lu, anorm, norm="1")
    if rcond < RCOND_FAIL:
        raise MatInvFailure("luc")
    logdet = float(np.sum(np.log(np.abs(u))))

    def solve(b):
        return sla.lu_solve((lu, piv), b, check_finite=False)

    return solve, logdet


def _norm1_blocked(matrix: np.ndarray) -> float:
    """max column sum of |a_ij| without an N x N temporary."""
    acc = np.zeros(matrix.shape[1])
    step = max(1, _BLOCK_ELEMS // matrix.shape[1])
    for i0 in range(0, matrix.shape[0], step):
        acc += np.abs(matrix[i0:i0 + step]).sum(axis=0)
    return float(acc.max())


def gp_log_likelihood_from_parts(r: np.ndarray, solve, logdet: float) -> float:
    """BGP:181-199: Clip[-0.5 (N log 2pi + LogDet + r.Inverse[r]), +-|$MachineLogZero|]."""
    val = -0.5 * (len(r) * LOG_TWO_PI + logdet + float(r @ solve(r)))
    lim = abs(MACHINE_LOG_ZERO)
    return float(min(max(val, -lim), lim))


def residual(kernel, theta, X, y, mean="zero", mean_fn=None):
    """BGP:300 Subtract[outputData, mean[#] /@ inputData].  mean_fn: optional function of ONE point (any m(x))."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    _, _, _, mu = split_theta(kernel, X.shape[1], theta, mean)
    if mean_fn is not None:
        return np.asarray(y, dtype=np.float64).ravel() - np.array([float(mean_fn(x)) for x in X])
    return np.asarray(y, dtype=np.float64).ravel() - mu


def log_likelihood(kernel: str, theta, X, y, mean: str = "zero", parts: bool = False, nugget_fn=None, mean_fn=None):
    """The closure assembled at BGP:297-305 (default branch) with Catch "MatInv" -> sentinel.
    parts=True additionally returns (logdet, quad, info).  nugget_fn / mean_fn: point-dependent forms (BGP:37, 300)."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    r = residual(kernel, theta, X, y, mean, mean_fn)
    try:
        K = covariance_matrix(kernel, theta, X, mean, nugget_fn)
        big = K.ndim == 2 and K.size > _BLOCK_ELEMS
        # (K is symmetric bit for bit, so its transpose view is the same matrix in the column-major order LAPACK
        #  factors in place)
        solve, logdet = matrix_inverse_and_det(K.T if big else K, overwrite=big)
        del K
    except MatInvFailure:
        return (MACHINE_LOG_ZERO, float("nan"), float("nan"), 1) if parts else MACHINE_LOG_ZERO
    ll = gp_log_likelihood_from_parts(r, solve, logdet)
    if parts:
        return ll, logdet, float(r @ solve(r)), 0
    return ll


def log_likelihood_mvn(kernel: str, theta, X, y, mean: str = "zero") -> float:
    """BGP:273-292, the ``Automatic`` branch: LogLikelihood[MultinormalDistribution[m, K], {y}]."""
    from scipy.stats import multivariate_normal
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    _, _, _, mu = split_theta(kernel, X.shape[1], theta, mean)
    K = covariance_matrix(kernel, theta, X, mean)
    if K.ndim == 1:
        K = np.diag(K)
    return float(multivariate_normal(mean=np.full(X.shape[0], mu), cov=K, allow_singular=False).logpdf(
        np.asarray(y, dtype=np.float64).ravel()))


def log_likelihood_grad(kernel: str, theta, X, y, mean: str = "zero") -> np.ndarray:
    """d/dtheta of BGP:190-196 by the standard identity  1/2 tr((alpha alpha^T - K^-1) dK/dtheta),
    alpha = K^-1 r  (no reference counterpart: the reference optimises with NMaximize and no
    gradients, LA:177-238).  Dense numpy, small N only."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    theta = np.asarray(theta, dtype=np.float64)
    n, d = X.shape
    ell, sf, sn, mu = split_theta(kernel, d, theta, mean)
    nl = n_lengthscales(kernel, d)
    K = covariance_matrix(kernel, theta, X, mean)
    Kinv = np.linalg.inv(K)
    r = np.asarray(y, dtype=np.float64).ravel() - mu
    alpha = Kinv @ r
    Wm = np.outer(alpha, alpha) - Kinv
    U2 = ((X[:, None, :] - X[None, :, :]) / ell) ** 2             # u_d^2, n x n x d
    r2 = U2.sum(axis=2)
    if kernel in ("se", "se_ard"):
        kpart = sf * sf * np.exp(-0.5 * r2)
        fac = kpart                                               # dK/dl_d = fac * u_d^2 / l_d
    else:
        s5 = np.sqrt(5.0 * r2)
        kpart = sf * sf * (1.0 + s5 + 5.0 / 3.0 * r2) * np.exp(-s5)
        fac = sf * sf * (5.0 / 3.0) * (1.0 + s5) * np.exp(-s5)
    grad = []
    if nl == 1:
        grad.append(0.5 * np.sum(Wm * fac * r2) / ell[0])
    else:
        for j in range(d):
            grad.append(0.5 * np.sum(Wm * fac * U2[:, :, j]) / ell[j])
    grad.append(0.5 * np.sum(Wm * kpart) * 2.0 / sf)
    grad.append(0.5 * np.trace(Wm) * 2.0 * sn)
    if mean == "const":
        grad.append(float(alpha.sum()))
    return np.array(grad)


# --------------------------------------------------------------------------------------
# Prediction: BGP:91-124 compiledKandKappa, BGP:396-422 predictFromGaussianProcessInternal
# --------------------------------------------------------------------------------------
def k_and_kappa(kernel: str, theta, X, Xs, mean: str = "zero", nugget_fn=None):
    """k: N x M (rows = train, cols = test; BGP:100-109), kappa_j = k(x*_j,x*_j)+nugget[x*_j] (BGP:110-115)."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    Xs = np.atleast_2d(np.asarray(Xs, dtype=np.float64))
    ell, sf, sn, _ = split_theta(kernel, X.shape[1], theta, mean)
    k = kernel_matrix(kernel, ell, sf, X, Xs)
    nug = np.full(Xs.shape[0], sn * sn) if nugget_fn is None else np.array([float(nugget_fn(x)) for x in Xs])
    kappa = (0.0 if kernel == "null" else sf * sf) + nug
    return k, kappa


def predict_internal(kernel: str, theta, X, y, Xs, mean: str = "zero", nugget_fn=None, mean_fn=None):
    """BGP:396-422: mu* = m(X*) + (K^-1 r).k (407-412); sigma* = Sqrt[kappa - Total[k * K^-1 k]]
    (414-417).  Returns (mu, sigma).  The variance includes the test-point nugget (BGP:113).
    nugget_fn / mean_fn: point-dependent nugget[x] / meanFunction[x] (BGP:37, 113, 300, 408)."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    _, _, _, mu0 = split_theta(kernel, X.shape[1], theta, mean)
    solve, _ = matrix_inverse_and_det(covariance_matrix(kernel, theta, X, mean, nugget_fn))
    k, kappa = k_and_kappa(kernel, theta, X, Xs, mean, nugget_fn)
    r = residual(kernel, theta, X, y, mean, mean_fn)
    if mean_fn is not None:
        mu0 = np.array([float(mean_fn(x)) for x in np.atleast_2d(np.asarray(Xs, dtype=np.float64))])
    mu = mu0 + solve(r) @ k
    var = kappa - np.sum(k * solve(k), axis=0)
    with np.errstate(invalid="ignore"):
        return mu, np.sqrt(var)


def predict_mixture(kernel: str, samples: np.ndarray, weights: np.ndarray, X, y, Xs, mean="zero"):
    """BGP:343-376: one Normal per posterior sample, mixed with the CrudePosteriorWeights
    (BGP:351-354).  Returns (weights[S], mu[S,M], sigma[S,M]) -- the content of the
    Association[x* -> MixtureDistribution[weights, {NormalDistribution..}]]."""
    mus, sds = [], []
    for th in np.atleast_2d(samples):
        m, s = predict_internal(kernel, th, X, y, Xs, mean)
        mus.append(m)
        sds.append(s)
    return np.asarray(weights, dtype=np.float64), np.array(mus), np.array(sds)


def mixture_moments(weights, mu, sigma):
    """Mean / variance of MixtureDistribution[w, Normal(mu_s, sigma_s)] per test point."""
    w = np.asarray(weights, dtype=np.float64)
    w = w / w.sum()
    m = w @ mu
    v = w @ (sigma ** 2 + mu ** 2) - m ** 2
    return m, v


# --------------------------------------------------------------------------------------
# BU:318-356 log-space helpers (used by the nested-sampling "next" row)
# --------------------------------------------------------------------------------------
def log_sum_exp(v) -> float:
    """BU:318-334 logSumExp: ignores -inf entries; empty / all -inf -> -inf."""
    v = np.asarray(v, dtype=np.float64).ravel()
    v = v[np.isfinite(v) | (v == np.inf)]
    if v.size == 0:
        return -np.inf
    m = np.max(v)
    return float(m + np.log(np.sum(np.exp(v - m))))


def log_add(a: float, b: float) -> float:
    """BU:336-346 logAdd."""
    hi, lo = (a, b) if a >= b else (b, a)
    if lo == -np.inf:
        return hi
    return hi + math.log1p(math.exp(lo - hi))


def log_subtract(a: float, b: float) -> float:
    """BU:348-356 logSubtract: log(e^a - e^b), a >= b."""
    if b == -np.inf:
        return a
    return a + math.log1p(-math.exp(b - a))

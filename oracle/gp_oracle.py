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

KERNELS = ("se", "se_ard", "matern52", "matern52_ard", "null", "matern32", "matern32_ard", "rq", "rq_ard")


def parse_kernel(kernel: str):
    """'term [(+|*) term] [+const]' -> ([term names], op in (None, '+', '*'), has_offset).  The reference takes any
    kernel[p, q] (BGP:32; its own example is `#2 + Exp[-(pt1-pt2)^2/#1^2]`, BGP:16): sums / products of the named
    families with an optional additive constant are the composed forms the build covers."""
    key = kernel.replace(" ", "").lower()
    offset = key.endswith("+const")
    if offset:
        key = key[:-len("+const")]
    for op in ("+", "*"):
        if op in key:
            a, b = key.split(op, 1)
            terms = [a, b]
            break
    else:
        terms, op = [key], None
    for t in terms:
        if t not in KERNELS or (t == "null" and (len(terms) > 1 or offset)):
            raise ValueError(kernel)
    return terms, op, offset


def _term_params(term: str, d: int) -> int:
    """hyper-parameters of one term: its length scales, alpha for the rational quadratic, sf"""
    return (1 if term in ("se", "matern52", "matern32", "rq") else d) + (1 if term.startswith("rq") else 0) + 1



class MatInvFailure(Exception):
    """Stands for Throw[$MachineLogZero, "MatInv"] (BGP:131-135)."""


# --------------------------------------------------------------------------------------
# Named kernels.  theta layout: (l_1..l_nl, sigma_f, sigma_n [, mu])
# --------------------------------------------------------------------------------------
def n_lengthscales(kernel: str, d: int) -> int:
    if kernel in ("se", "matern52", "matern32", "rq"):
        return 1
    if kernel in ("se_ard", "matern52_ard", "matern32_ard", "rq_ard"):
        return d
    if kernel == "null":
        return 0
    raise ValueError(kernel)


def is_custom(kernel) -> bool:
    """A covariance function given as a Python function instead of a name: any object with `fn(A, B, p)` (numpy, broadcasting
    over point arrays [.., d]) and `nparams`; theta = [p_0 .. p_{nparams-1}, sn (, mu)].  Stands for the reference's arbitrary
    `kernel @@ points[[{i,j}]]` (BGP:29-33, 100-109)."""
    return hasattr(kernel, "fn") and hasattr(kernel, "nparams") and not isinstance(kernel, str)


def custom_kernel_matrix(kernel, theta, A: np.ndarray, B: np.ndarray) -> np.ndarray:
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    B = np.atleast_2d(np.asarray(B, dtype=np.float64))
    p = np.asarray(theta, dtype=np.float64)[:kernel.nparams]
    rows = max(1, _BLOCK_ELEMS // max(1, B.shape[0] * A.shape[1]))
    out = np.empty((A.shape[0], B.shape[0]))
    for r0 in range(0, A.shape[0], rows):
        out[r0:r0 + rows] = kernel.fn(A[r0:r0 + rows, None, :], B[None, :, :], p)
    return out


def n_params(kernel: str, d: int, mean: str = "zero") -> int:
    if is_custom(kernel):
        return kernel.nparams + 1 + (1 if mean == "const" else 0)
    # null kernel: theta = (sigma_n [, mu]); general: [term 1: l.., (alpha), sf] [term 2 ..] [c] sn [mu]
    terms, _, offset = parse_kernel(kernel)
    base = 1 if kernel == "null" else sum(_term_params(t, d) for t in terms) + (1 if offset else 0) + 1
    return base + (1 if mean == "const" else 0)


def split_general(kernel: str, d: int, theta, mean: str = "zero"):
    """theta -> ([(term, ell[d], alpha, sf), ..], op, c, sn, mu) for any kernel of the grammar of parse_kernel."""
    theta = np.asarray(theta, dtype=np.float64)
    terms, op, offset = parse_kernel(kernel)
    out, o = [], 0
    for t in terms:
        nl = n_lengthscales(t, d)
        ell = np.broadcast_to(theta[o:o + nl], (d,)) if nl == 1 else theta[o:o + nl]
        o += nl
        alpha = None
        if t.startswith("rq"):
            alpha = float(theta[o])
            o += 1
        out.append((t, np.asarray(ell, dtype=np.float64), alpha, float(theta[o])))
        o += 1
    c = 0.0
    if offset:
        c = float(theta[o])
        o += 1
    sn = float(theta[o])
    o += 1
    mu = float(theta[o]) if mean == "const" else 0.0
    return out, op, c, sn, mu


def split_theta(kernel: str, d: int, theta, mean: str = "zero"):
    theta = np.asarray(theta, dtype=np.float64)
    if is_custom(kernel):
        m = kernel.nparams
        return np.ones(d), 1.0, float(theta[m]), (float(theta[m + 1]) if mean == "const" else 0.0)
    if kernel not in ("se", "se_ard", "matern52", "matern52_ard", "null"):      # general form: callers use split_general
        terms, _, _, sn, mu = split_general(kernel, d, theta, mean)
        return terms[0][1], terms[0][3], sn, mu
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
    if A.shape[0] * B.shape[0] > _BLOCK_ELEMS or (BUILD_THREADS > 1 and A.shape[0] >= 512):
        # large problems (golden scalars at N = 16384 / 32768, the bench's CPU baseline): the same arithmetic,
        # element for element, in row blocks so that the temporaries stay small next to the result
        out = np.empty((A.shape[0], B.shape[0]))
        step = max(1, _BLOCK_ELEMS // B.shape[0])
        if BUILD_THREADS > 1:
            # bench.py's cpu_baseline only: the row blocks on a thread pool (numpy's element-wise loops release the GIL).
            # Every element goes through the same operations in the same order -- the matrix is bit-identical.
            from concurrent.futures import ThreadPoolExecutor
            step = max(1, min(step, -(-A.shape[0] // (4 * BUILD_THREADS))))

            def block(i0):
                out[i0:i0 + step] = _kernel_block(kernel, sf, A[i0:i0 + step], B)
            with ThreadPoolExecutor(BUILD_THREADS) as pool:
                list(pool.map(block, range(0, A.shape[0], step)))
            return out
        for i0 in range(0, A.shape[0], step):
            out[i0:i0 + step] = _kernel_block(kernel, sf, A[i0:i0 + step], B)
        return out
    return _kernel_block(kernel, sf, A, B)


_BLOCK_ELEMS = 1 << 26          # 512 MiB of fp64 per temporary
BUILD_THREADS = 1               # > 1: kernel_matrix's row blocks run on a thread pool (set by bench.py's CPU baseline only)


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
    if kernel in ("matern32", "matern32_ard"):
        s3 = math.sqrt(3.0) * np.sqrt(r2)
        return (sf * sf) * (1.0 + s3) * np.exp(-s3)
    if kernel == "null":
        return np.zeros_like(r2)
    raise ValueError(kernel)


def general_kernel_matrix(kernel: str, theta, A: np.ndarray, B: np.ndarray, mean: str = "zero") -> np.ndarray:
    """k(a_i, b_j) = [c +] k1 [(+|*) k2] for the composed forms (and the plain families with their own theta layout)."""
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    B = np.atleast_2d(np.asarray(B, dtype=np.float64))
    terms, op, c, _, _ = split_general(kernel, A.shape[1], theta, mean)
    mats = []
    for t, ell, alpha, sf in terms:
        if t.startswith("rq"):
            r2 = np.zeros((A.shape[0], B.shape[0]))
            for j in range(A.shape[1]):
                diff = A[:, j][:, None] / ell[j] - B[:, j][None, :] / ell[j]
                r2 += diff * diff
            mats.append((sf * sf) * np.power(1.0 + r2 / (2.0 * alpha), -alpha))
        else:
            mats.append(kernel_matrix(t, ell, sf, A, B))
    K = mats[0] if op is None else (mats[0] + mats[1] if op == "+" else mats[0] * mats[1])
    return K + c


def covariance_matrix(kernel: str, theta, X: np.ndarray, mean: str = "zero", nugget_fn=None):
    """BGP:27-43 covarianceMatrix: K_ij = k(x_i,x_j) + delta_ij nugget(x_i).
    Null kernel (BGP:25-27): returns the diagonal *vector* nugget /@ points.
    nugget_fn: optional function of ONE point (`nugget[points[[i]]]`, BGP:37) returning a variance; None = the constant
    Function[sn^2]."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    ell, sf, sn, _ = split_theta(kernel, X.shape[1], theta, mean)
    nug = np.full(X.shape[0], sn * sn) if nugget_fn is None else np.array([float(nugget_fn(x)) for x in X])
    if is_custom(kernel):
        K = custom_kernel_matrix(kernel, theta, X, X)
        K[np.diag_indices_from(K)] += nug
        return K
    if kernel == "null":
        return nug
    if kernel in ("se", "se_ard", "matern52", "matern52_ard"):
        K = kernel_matrix(kernel, ell, sf, X, X)
    else:
        K = general_kernel_matrix(kernel, theta, X, X, mean)
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
    if kernel not in ("se", "se_ard", "matern52", "matern52_ard"):
        # general forms: 4th-order central differences of the oracle's own log-likelihood (small N only)
        grad = np.zeros(len(theta))
        for i in range(len(theta)):
            hstep = 1e-3 * max(abs(theta[i]), 0.1)
            f = lambda t: log_likelihood(kernel, np.concatenate([theta[:i], [theta[i] + t], theta[i + 1:]]), X, y, mean)  # noqa: E731
            grad[i] = (-f(2 * hstep) + 8 * f(hstep) - 8 * f(-hstep) + f(-2 * hstep)) / (12 * hstep)
        return grad
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
    nug = np.full(Xs.shape[0], sn * sn) if nugget_fn is None else np.array([float(nugget_fn(x)) for x in Xs])
    if is_custom(kernel):
        p = np.asarray(theta, dtype=np.float64)[:kernel.nparams]
        return custom_kernel_matrix(kernel, theta, X, Xs), np.asarray(kernel.fn(Xs, Xs, p), dtype=np.float64) + nug
    if kernel in ("se", "se_ard", "matern52", "matern52_ard", "null"):
        k = kernel_matrix(kernel, ell, sf, X, Xs)
        kappa = (0.0 if kernel == "null" else sf * sf) + nug
    else:
        k = general_kernel_matrix(kernel, theta, X, Xs, mean)
        kappa = np.array([general_kernel_matrix(kernel, theta, x[None, :], x[None, :], mean)[0, 0] for x in Xs]) + nug
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

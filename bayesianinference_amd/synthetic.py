"""Deterministic synthetic GP data, counter-based splitmix64.

The reference ships no data sets or benchmarks, so the bench and the
parity tests use this generator.  It is counter based -- output ``idx`` of stream ``s`` is
``mix64(seed + s*STREAM + (idx+1)*GOLDEN)`` -- so it vectorises in numpy and any slice can
be regenerated independently on every rank (multi-GPU sharding needs no broadcast of X).

    X[i, j] = 2u - 1                       (uniform [-1, 1)^d, row-major N x d)
    y[i]    = sin(2 * sum_j X[i, j]/(1+j)) + 0.1 * g_i,   g ~ N(0,1) (Box-Muller)
"""
from __future__ import annotations

import numpy as np

SEED = 20250905
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_STREAM = np.uint64(0xD1B54A32D192ED03)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

STREAM_X = 0
STREAM_NOISE = 1
STREAM_THETA = 2
STREAM_TEST = 3


def _mix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def uniform(stream: int, start: int, count: int, seed: int = SEED) -> np.ndarray:
    """``count`` doubles in [0,1) at positions start..start+count-1 of ``stream``."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        base = np.uint64(seed) + np.uint64(stream) * _STREAM
        bits = _mix64(base + idx * _GOLDEN)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(stream: int, start: int, count: int, seed: int = SEED) -> np.ndarray:
    """Box-Muller normals; element i uses uniforms 2i and 2i+1 of the stream."""
    u = uniform(stream, 2 * start, 2 * count, seed)
    u1 = 1.0 - u[0::2]          # (0, 1]
    u2 = u[1::2]
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def make_inputs(n: int, d: int, seed: int = SEED, stream: int = STREAM_X,
                row0: int = 0) -> np.ndarray:
    """Rows row0..row0+n-1 of the (infinite) synthetic design matrix, N x d row-major."""
    u = uniform(stream, row0 * d, n * d, seed)
    return (2.0 * u - 1.0).reshape(n, d)


def make_outputs(X: np.ndarray, seed: int = SEED, row0: int = 0) -> np.ndarray:
    n, d = X.shape
    w = 1.0 / (1.0 + np.arange(d))
    g = normal(STREAM_NOISE, row0, n, seed)
    return np.sin(2.0 * (X @ w)) + 0.1 * g


def make_dataset(n: int, d: int, seed: int = SEED):
    X = make_inputs(n, d, seed)
    return X, make_outputs(X, seed)


def make_test_points(m: int, d: int, seed: int = SEED) -> np.ndarray:
    return make_inputs(m, d, seed, stream=STREAM_TEST)


def default_theta(kernel: str, d: int, dtype: str = "f64") -> np.ndarray:
    """Timing hyper-parameters of SURVEY.md §8d: l=1 (0.3 for d=1), sf=1, sn=0.1 (0.3 fp32)."""
    sn = 0.1 if dtype == "f64" else 0.3
    ell = 0.3 if d == 1 else 1.0
    if kernel in ("se", "matern52"):
        return np.array([ell, 1.0, sn])
    return np.concatenate([np.full(d, ell), [1.0, sn]])


def theta_batch(b: int, kernel: str, d: int, seed: int = SEED) -> np.ndarray:
    """cfg-4 style batch: log-uniform l in [0.1,10], sf in [0.1,10], sn in [0.01,1]."""
    nl = 1 if kernel in ("se", "matern52") else d
    p = nl + 2
    u = uniform(STREAM_THETA, 0, b * p, seed).reshape(b, p)
    lo = np.concatenate([np.full(nl, 0.1), [0.1, 0.01]])
    hi = np.concatenate([np.full(nl, 10.0), [10.0, 1.0]])
    return np.exp(np.log(lo) + u * (np.log(hi) - np.log(lo)))


def make_clustered(n: int, d: int, clusters: int = 100, spread: float = 1e-4, noise: float = 0.1, seed: int = SEED):
    """Near-duplicate inputs: ``clusters`` centres uniform in [-0.99, 0.99]^d, every point within ``spread`` of one of
    them, plus the two corners (-1,..,-1), (1,..,1) so that the half range is exactly 1.  The adversarial case for a
    kernel build that forms r^2 = |a|^2 + |b|^2 - 2 a.b: short length scales make the norms large while the
    near-duplicates make K as ill-conditioned as the nugget allows.
    y = sin(2 sum_j x_j / (1 + j)) + noise * g."""
    cen = 0.99 * (2.0 * uniform(STREAM_X, 0, clusters * d, seed + 17).reshape(clusters, d) - 1.0)
    which = (uniform(STREAM_X, 0, n, seed + 18) * clusters).astype(np.int64) % clusters
    off = spread * (2.0 * uniform(STREAM_X, 0, n * d, seed + 19).reshape(n, d) - 1.0)
    X = cen[which] + off
    X[0] = -1.0
    X[1] = 1.0
    w = 1.0 / (1.0 + np.arange(d))
    y = np.sin(2.0 * (X @ w)) + noise * normal(STREAM_NOISE, 0, n, seed + 20)
    return X, y

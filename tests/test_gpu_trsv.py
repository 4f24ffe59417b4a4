"""Single-vector substitutions (csrc/gp_trsv.h, trsv_dataflow_kernel): "Inverse"[vector] = K^-1 b of the reference's
"InverseCovarianceFunction" association (BGP:130-141, 194, 407-412) with 1 .. 4 right-hand sides -- one launch per triangle
that streams the factor once, hand-offs by sentinel polling -- against numpy's solve on the oracle's K (BGP:29-43) at 1e-8,
against the GEMM-shaped substitution (option trsv = 0), run to run (bit-identical: the row sums are chains, not atomics),
for every shape of the task list (1, 2, 3 tile columns: no tile tasks / chain only) and in fp32."""
import numpy as np
import pytest

from bayesianinference_amd import _lib, synthetic as syn
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kernel,d,n,nrhs", [("se_ard", 3, 100, 1), ("se_ard", 3, 256, 2), ("matern52", 2, 300, 1), ("se_ard", 3, 384, 4),
                                             ("se", 1, 640, 3), ("se_ard", 8, 1500, 1), ("se_ard", 8, 3000, 4), ("matern52_ard", 5, 5000, 1),
                                             ("se_ard", 8, 8192, 1), ("se_ard", 8, 8192, 4)])
def test_solve_few_vectors_matches_numpy_and_the_gemm_substitution(kernel, d, n, nrhs):
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta(kernel, d)
    th[-1] = 0.3
    K = orc.covariance_matrix(kernel, th, X)
    rng = np.random.default_rng(n + nrhs)
    B = rng.standard_normal((n, nrhs))
    want = np.linalg.solve(K, B)
    h = _lib.Handle(X, y, kernel)
    assert h.fit(th) == 0
    assert h.get_option("trsv") == 1
    got = h.solve(B if nrhs > 1 else B[:, 0]).reshape(n, -1)
    again = h.solve(B if nrhs > 1 else B[:, 0]).reshape(n, -1)
    h.set_option("trsv", 0)
    old = h.solve(B if nrhs > 1 else B[:, 0]).reshape(n, -1)
    h.close()
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-9 * scale)
    np.testing.assert_allclose(got, old, rtol=1e-9, atol=1e-10 * scale)
    assert np.array_equal(got, again)
    # K K^-1 b = b
    assert np.abs(K @ got - B).max() <= 1e-9 * max(1.0, np.abs(B).max()) * n


def test_solve_after_every_kind_of_fit_and_in_fp32():
    """The 128-block inverses the kernel reads exist after a single-launch (64-tile) fit, a 128-tile dataflow fit and the
    look-ahead schedule alike; fp32 against the fp64 oracle at fp32 accuracy."""
    n, d = 2100, 4
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d)
    th[-1] = 0.3
    K = orc.covariance_matrix("se_ard", th, X)
    b = np.random.default_rng(0).standard_normal(n)
    want = np.linalg.solve(K, b)
    for opts in ({}, {"dataflow": 0}, {"dataflow_fine_nt": 0}, {"dataflow": 0, "lookahead": 0}):
        h = _lib.Handle(X, y, "se_ard")
        for k, v in opts.items():
            h.set_option(k, v)
        assert h.fit(th) == 0
        got = h.solve(b)
        h.close()
        np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-9 * np.abs(want).max(), err_msg=str(opts))
    th32 = syn.default_theta("se_ard", d, dtype="f32")
    K32 = orc.covariance_matrix("se_ard", th32, X.astype(np.float32).astype(np.float64))
    want32 = np.linalg.solve(K32, b)
    h = _lib.Handle(X, y, "se_ard", dtype=32)
    assert h.fit(th32) == 0
    got32 = h.solve(b)
    h.set_option("trsv", 0)
    old32 = h.solve(b)
    h.close()
    assert np.abs(got32 - want32).max() <= 2e-3 * np.abs(want32).max()
    assert np.abs(got32 - old32).max() <= 2e-3 * np.abs(want32).max()


def test_gradient_alpha_through_the_backward_launch():
    """alpha = K^-1 r of the gradient (queue_alpha) on the look-ahead schedule is one backward launch of the same kernel: the
    gradient against the oracle (BGP:181-199 differentiated; LA:177-238 is its caller) with and without it."""
    n, d = 2600, 3
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d)
    want = orc.log_likelihood_grad("se_ard", th, X, y)
    res = {}
    for trsv in (1, 0):
        h = _lib.Handle(X, y, "se_ard")
        h.set_option("grad_potri", 0)          # (rows of K^-1 by substitution; alpha by queue_alpha)
        h.set_option("trsv", trsv)
        ll, g, info = h.loglik_grad(th)
        h.close()
        assert info == 0
        np.testing.assert_allclose(g, want, rtol=1e-7, atol=1e-7 * np.abs(want).max())
        res[trsv] = g
    np.testing.assert_allclose(res[1], res[0], rtol=1e-9, atol=1e-9 * np.abs(want).max())


def test_up_to_sixteen_vectors_in_batches_at_large_n():
    """From N = 12288 on, 5 .. 16 right-hand sides go through the single-vector launches in batches of four (there the
    GEMM-shaped substitution is hundreds of launches: five vectors at N = 16384 14.6 -> 2.9 ms): the same solutions as the
    GEMM-shaped path (itself held to the oracle elsewhere), ragged N, and K x = b on sampled rows (BGP:29-43 for the rows)."""
    n, d, nrhs = 12400, 4, 6
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d)
    th[-1] = 0.3
    B = np.random.default_rng(5).standard_normal((n, nrhs))
    h = _lib.Handle(X, y, "se_ard")
    assert h.fit(th) == 0
    got = h.solve(B)
    h.set_option("trsv", 0)
    old = h.solve(B)
    h.close()
    scale = np.abs(old).max()
    np.testing.assert_allclose(got, old, rtol=1e-9, atol=1e-10 * scale)
    rows = np.array([0, 1, 127, 128, 5000, 12287, 12288, n - 1])
    Krows = orc.kernel_matrix("se_ard", th[:d], th[d], X[rows], X)
    Krows[np.arange(len(rows)), rows] += th[-1] ** 2
    assert np.abs(Krows @ got - B[rows]).max() <= 1e-8 * n

"""ANY covariance function on the device (gphip_create_custom): the reference evaluates an arbitrary pure function
`kernel @@ points[[{i,j}]]` (BGP:29-33; cross form BGP:100-109; kappa BGP:110-115).  The caller's function arrives as source
text, is compiled at run time (hiprtc) into the library's own kernel build, and everything downstream -- factorisation,
log-likelihood, prediction, posterior-sample mixtures -- is the same code as for the named kernels.  Parity: against the CPU
oracle evaluating the same function in numpy (1e-8 fp64, 1e-3 fp32), and against the named-kernel handle where one exists."""
import numpy as np
import pytest

from bayesianinference_amd import _lib, synthetic as syn
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

SE_ARD_BODY = "T s = 0; for (int k = 0; k < D; ++k) { const T u = (X(k) - Y(k)) / P(k); s += u * u; } return P(D) * P(D) * exp((T)-0.5 * s);"


def se_ard_fn(A, B, p):
    d = A.shape[-1]
    return p[d] ** 2 * np.exp(-0.5 * (((A - B) / p[:d]) ** 2).sum(-1))


# non-stationary: squared exponential x (1 + c x_0 x'_0)  -- k(x, x) depends on the point
NONSTAT_BODY = ("T s = 0; for (int k = 0; k < D; ++k) { const T u = X(k) - Y(k); s += u * u; } "
                "return P(1) * P(1) * exp((T)-0.5 * s / (P(0) * P(0))) * ((T)1 + P(2) * P(2) * X(0) * Y(0));")


def nonstat_fn(A, B, p):
    return p[1] ** 2 * np.exp(-0.5 * ((A - B) ** 2).sum(-1) / p[0] ** 2) * (1.0 + p[2] ** 2 * A[..., 0] * B[..., 0])


# periodic (MacKay), the way Mathematica's CForm would print it for d = 1
PERIODIC_BODY = "return Power(P(2),2)*Exp((-2*Power(Sin((Pi*(X(0) - Y(0)))/P(1)),2))/Power(P(0),2));"


def periodic_fn(A, B, p):
    return p[2] ** 2 * np.exp(-2.0 * np.sin(np.pi * (A[..., 0] - B[..., 0]) / p[1]) ** 2 / p[0] ** 2)


def close(a, b, n, tol=1e-8):
    return abs(a - b) <= tol * max(1.0, abs(b), n)


@pytest.mark.parametrize("n,d", [(700, 3), (300, 8), (1500, 2), (260, 40)])
def test_se_ard_as_source_text_matches_named_kernel_and_oracle(n, d):
    X, y = syn.make_dataset(n, d)
    Xs = syn.make_test_points(100, d)
    th = syn.default_theta("se_ard", d)
    ck = _lib.CustomKernel(SE_ARD_BODY, d + 1, fn=se_ard_fn)
    h = _lib.Handle(X, y, ck)
    ref = _lib.Handle(X, y, "se_ard")
    assert h.p == d + 2
    ll, ld, qd, info = h.loglik_parts(th)
    l0, ld0, qd0, _ = ref.loglik_parts(th)
    assert info == 0 and close(ll, l0, n, 1e-10) and close(ld, ld0, n, 1e-10) and close(qd, qd0, n, 1e-9)
    want = orc.log_likelihood(ck, th, X, y, parts=True)
    assert close(ll, want[0], n) and close(ld, want[1], n) and close(qd, want[2], n)
    assert h.fit(th) == 0 and ref.fit(th) == 0
    mu, var = h.predict(Xs)
    mu0, var0 = ref.predict(Xs)
    np.testing.assert_allclose(mu, mu0, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(var, var0, rtol=1e-7, atol=1e-12)
    # batch of thetas, each slot with its own parameters
    Th = np.array([th * (1.0 + 0.03 * k) for k in range(5)])
    lb, ib = h.loglik_batch(Th)
    lb0, _ = ref.loglik_batch(Th)
    assert (ib == 0).all()
    np.testing.assert_allclose(lb, lb0, rtol=1e-10, atol=1e-8)
    h.close(); ref.close()


@pytest.mark.parametrize("body,fn,npar,d,theta", [
    (NONSTAT_BODY, nonstat_fn, 3, 2, [0.9, 1.2, 0.7, 0.15]),
    (NONSTAT_BODY, nonstat_fn, 3, 5, [1.6, 0.8, 0.4, 0.2]),
    (PERIODIC_BODY, periodic_fn, 3, 1, [1.1, 2.3, 1.4, 0.1]),
])
def test_arbitrary_covariance_functions_against_the_oracle(body, fn, npar, d, theta):
    n = 900
    X, y = syn.make_dataset(n, d)
    Xs = syn.make_test_points(150, d)
    th = np.array(theta)
    ck = _lib.CustomKernel(body, npar, fn=fn)
    h = _lib.Handle(X, y, ck)
    ll, ld, qd, info = h.loglik_parts(th)
    want = orc.log_likelihood(ck, th, X, y, parts=True)
    assert info == 0 and want[3] == 0
    assert close(ll, want[0], n) and close(ld, want[1], n) and close(qd, want[2], n)
    assert h.fit(th) == 0
    mu, var = h.predict(Xs)
    mo, so = orc.predict_internal(ck, th, X, y, Xs)
    np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.sqrt(var), so, rtol=1e-7)       # (k(x*, x*) is a function of the test point here)
    # the covariance matrix itself, entry by entry; the cross form with kappa = k(x*, x*) + nugget (BGP:100-115)
    K = h.covariance(th)
    np.testing.assert_allclose(K, orc.covariance_matrix(ck, th, X), rtol=1e-12, atol=1e-14)
    kx, kap = h.cross_covariance(th, Xs)
    ko, kapo = orc.k_and_kappa(ck, th, X, Xs)
    np.testing.assert_allclose(kx, ko, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(kap, kapo, rtol=1e-12)
    # constant mean, fp32, verdicts
    hc = _lib.Handle(X, y, ck, mean="const")
    thc = np.append(th, 0.2)
    assert close(hc.loglik(thc)[0], orc.log_likelihood(ck, thc, X, y, "const"), n)
    h32 = _lib.Handle(X, y, ck, dtype=32)
    l32, i32 = h32.loglik(th)
    assert i32 == 0 and abs(l32 - want[0]) <= 1e-3 * max(1.0, abs(want[0]), n)
    bad = th.copy(); bad[0] = np.nan
    assert h.loglik(bad)[1] == _lib.INFO_NAN and h.loglik(th)[1] == 0
    Xd = X.copy(); Xd[n // 2] = Xd[3]
    hd = _lib.Handle(Xd, y, ck)
    zero_nug = th.copy(); zero_nug[-1] = 0.0
    assert hd.loglik(zero_nug)[1] == _lib.INFO_NOT_SPD and hd.loglik(th)[1] == 0
    # gradient: one factorisation, the function instantiated with dual numbers (test_gradient_* below look closer)
    gl, gg, gi = h.loglik_grad(th)
    assert gi == 0 and close(gl, ll, n, 1e-12) and h.get_option("grad_analytic") == 1
    eps = 1e-5
    for k in range(len(th)):
        tp, tm = th.copy(), th.copy()
        tp[k] += eps; tm[k] -= eps
        fd = (orc.log_likelihood(ck, tp, X, y) - orc.log_likelihood(ck, tm, X, y)) / (2 * eps)
        assert abs(gg[k] - fd) <= 2e-5 * max(1.0, abs(fd)), (k, gg[k], fd)
    for x in (h, hc, h32, hd):
        x.close()


def test_posterior_sample_mixture_and_errors():
    n, d = 400, 2
    X, y = syn.make_dataset(n, d)
    Xs = syn.make_test_points(40, d)
    ck = _lib.CustomKernel(NONSTAT_BODY, 3, fn=nonstat_fn)
    h = _lib.Handle(X, y, ck)
    samples = np.array([[0.9, 1.2, 0.7, 0.15], [1.0, 1.1, 0.5, 0.2], [0.8, 1.3, 0.9, 0.12]])
    mu, var, info = h.predict_samples(samples, Xs)
    assert (info == 0).all()
    for s in range(3):
        mo, so = orc.predict_internal(ck, samples[s], X, y, Xs)
        np.testing.assert_allclose(mu[s], mo, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(np.sqrt(var[s]), so, rtol=1e-7)
    h.close()
    with pytest.raises(_lib.GphipError) as e:            # the compiler's log comes back with the error
        _lib.Handle(X, y, _lib.CustomKernel("return P(0) * no_such_symbol;", 1))
    assert e.value.status == 1 and "no_such_symbol" in str(e.value)


@pytest.mark.parametrize("world,panel", [(2, 2), (3, 1)])
def test_function_valued_kernel_on_a_multi_device_handle(world, panel):
    """A device list makes ONE multi-device handle (virtual ranks on the one GPU here): every member compiles the function for
    itself; a sharded evaluation builds each rank's own panels with it, a sharded fit predicts with per-point prior variances."""
    n, d = 1300, 2
    X, y = syn.make_dataset(n, d)
    Xs = syn.make_test_points(60, d)
    th = np.array([0.9, 1.2, 0.7, 0.15])
    ck = _lib.CustomKernel(NONSTAT_BODY, 3, fn=nonstat_fn)
    g = _lib.Handle(X, y, ck, device=[0] * world)
    g.set_option("shard_min_n", 0)
    g.set_option("panel", panel)
    ll, ld, qd, info = g.loglik_parts(th)
    want = orc.log_likelihood(ck, th, X, y, parts=True)
    assert info == 0 and close(ll, want[0], n) and close(ld, want[1], n) and close(qd, want[2], n)
    mo, so = orc.predict_internal(ck, th, X, y, Xs)
    for replicate in (0, 1):
        g.set_option("replicate_factor", replicate)
        assert g.fit(th) == 0
        mu, var = g.predict(Xs)
        np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(np.sqrt(var), so, rtol=1e-7)
    Th = np.array([th * (1.0 + 0.02 * k) for k in range(5)])         # thetas dealt to the members
    lb, ib = g.loglik_batch(Th)
    assert (ib == 0).all()
    for k in range(5):
        assert close(lb[k], orc.log_likelihood(ck, Th[k], X, y), n)
    g.close()


def test_host_mirror_object_with_a_function_valued_kernel(tmp_path):
    """defineGaussianProcess / predictFromGaussianProcess / persistence with a CustomKernel: the object a reference user builds
    with an arbitrary `kernel` pure function (BGP:228-330), its closure against the oracle, the saved file carrying the source text."""
    from bayesianinference_amd import gaussian_process as gp
    X, y = syn.make_dataset(300, 2)
    ck = _lib.CustomKernel(NONSTAT_BODY, 3, fn=nonstat_fn, name="se_times_linear")
    variables = [("l", 0.2, 5.0), ("sf", 0.2, 5.0), ("c", 0.0, 2.0), ("sn", 0.02, 1.0)]
    obj = gp.defineGaussianProcess((X, y), ck, variables=variables)
    assert not obj.failed and obj["GaussianProcessData"]["ModelFunctions"]["KernelFunction"] == ("se_times_linear", NONSTAT_BODY)
    th = np.array([0.9, 1.2, 0.7, 0.15])
    assert obj["LogLikelihoodFunction"](th) == pytest.approx(orc.log_likelihood(ck, th, X, y), rel=1e-8)
    K = obj["GaussianProcessData"]["ModelFunctions"]["CovarianceFunction"](th)
    np.testing.assert_allclose(K, orc.covariance_matrix(ck, th, X), rtol=1e-12, atol=1e-14)
    pts = syn.make_test_points(9, 2)
    samples = [{"Point": th, "CrudePosteriorWeight": 1.0, "CrudeLogPosteriorWeight": 0.0}]
    a = gp.predictFromGaussianProcess(obj.append({"Samples": samples}), pts)
    mo, so = orc.predict_internal(ck, th, X, y, pts)
    np.testing.assert_allclose(a["Mean"][0], mo, rtol=1e-7, atol=1e-9)
    path = str(tmp_path / "gp_custom.npz")
    gp.save_gaussian_process(obj.append({"Samples": samples}), path, theta=th)
    with pytest.raises(ValueError):                            # a checkpoint with kernel source is executable content
        gp.load_gaussian_process(path)
    obj2, th2 = gp.load_gaussian_process(path, trust_kernel_source=True)
    assert not obj2.failed and obj2["KernelName"].body == NONSTAT_BODY and obj2["KernelName"].nparams == 3
    b = gp.predictFromGaussianProcess(obj2, pts)
    np.testing.assert_allclose(a["Mean"], b["Mean"], rtol=1e-12)
    with pytest.raises(ValueError):                      # wrong number of variables for the function's parameter count
        gp.defineGaussianProcess((X, y), ck, variables=variables[:3])


def test_native_sampler_runs_on_a_function_valued_kernel():
    """gphip_nested_sampling drives batched likelihood calls of the handle: with SE-ARD given as source text it must reproduce,
    draw for draw (same seed), the run on the named-kernel handle."""
    X, y = syn.make_dataset(96, 1)
    ck = _lib.CustomKernel(SE_ARD_BODY, 2, fn=se_ard_fn)
    box = np.array([[0.05, 1.5], [0.2, 3.0], [0.03, 0.6]])
    opts = dict(pool=30, max_iterations=150, min_iterations=20, mc_steps=8, walkers=8, seed=7)
    h, ref = _lib.Handle(X, y, ck), _lib.Handle(X, y, "se_ard")
    a, b = h.nested_sampling(box, **opts), ref.nested_sampling(box, **opts)
    assert a["TotalSamples"] == b["TotalSamples"] and np.isfinite(a["CrudeLogEvidence"])
    np.testing.assert_allclose(a["Points"], b["Points"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(a["LogLikelihood"], b["LogLikelihood"], rtol=1e-9, atol=1e-8)
    assert abs(a["CrudeLogEvidence"] - b["CrudeLogEvidence"]) < 1e-6
    h.close(); ref.close()


def test_function_valued_kernel_under_every_factorisation_schedule():
    """N = 6000: the single dataflow launch (default), the multi-kernel look-ahead schedule with its dataflow tail (dataflow
    launch off for the whole matrix), fused dataflow panels, and a 3-rank sharded evaluation -- all around the run-time compiled
    kernel build, all against the oracle evaluating the same non-stationary function."""
    n, d = 6000, 3
    X, y = syn.make_dataset(n, d)
    th = np.array([0.8, 1.1, 0.6, 0.12])
    ck = _lib.CustomKernel(NONSTAT_BODY, 3, fn=nonstat_fn)
    want = orc.log_likelihood(ck, th, X, y, parts=True)
    assert want[3] == 0
    h = _lib.Handle(X, y, ck)
    for opts in ({}, {"dataflow_max_nt": 16}, {"dataflow_max_nt": 16, "panel_df": 1, "dataflow_tail": 16}, {"lookahead": 0, "dataflow": 0}):
        for k, v in opts.items():
            h.set_option(k, v)
        ll, ld, qd, info = h.loglik_parts(th)
        assert info == 0 and close(ll, want[0], n) and close(ld, want[1], n) and close(qd, want[2], n), opts
        for k in opts:
            h.set_option(k, {"dataflow_max_nt": 96, "panel_df": -1, "dataflow_tail": 64, "lookahead": 1, "dataflow": 1}[k])
    h.close()
    g = _lib.Handle(X, y, ck, device=[0, 0, 0])
    g.set_option("shard_min_n", 0)
    for mode in (0, 2):
        g.set_option("dist_panel_df", mode)
        ll, ld, qd, info = g.loglik_parts(th)
        assert info == 0 and close(ll, want[0], n) and close(ld, want[1], n) and close(qd, want[2], n), mode
    g.close()


def test_cform_names_available_to_a_function_body():
    """Every name Mathematica's CForm can emit for an elementary kernel expression resolves inside the generated source."""
    X, y = syn.make_dataset(200, 1)
    body = ("const T r = Abs(X(0) - Y(0)); "
            "return Power(P(0),2) * Exp(-r) * (1 + Tanh(Min(r, 2.0)) * 0) * Max(Cos(0 * r), 0.5) + 0 * (Sin(r) + Tan(r) + ArcTan(r) + "
            "Sinh(r) + Cosh(r) + Erf(r) + Erfc(r) + Log(1 + r) + Sqrt(r) + Pi + E);")
    ck = _lib.CustomKernel(body, 1, fn=lambda A, B, p: p[0] ** 2 * np.exp(-np.abs(A[..., 0] - B[..., 0])))
    h = _lib.Handle(X, y, ck)
    th = np.array([1.3, 0.2])
    ll, info = h.loglik(th)
    assert info == 0 and close(ll, orc.log_likelihood(ck, th, X, y), 200)
    h.close()


def _fd_grad(ck, th, X, y, mean="zero", eps=1e-5):
    g = np.zeros(len(th))
    for k in range(len(th)):
        tp, tm = th.copy(), th.copy()
        tp[k] += eps; tm[k] -= eps
        g[k] = (orc.log_likelihood(ck, tp, X, y, mean) - orc.log_likelihood(ck, tm, X, y, mean)) / (2 * eps)
    return g


@pytest.mark.parametrize("n,d", [(700, 3), (1500, 8), (260, 40)])
def test_gradient_by_dual_numbers_matches_the_named_kernel_and_the_oracle(n, d):
    """SE-ARD handed over as text: gphip_loglik_grad instantiates the text with forward-mode dual numbers (gp_dual.h) inside
    custom_grad_kernel -- ONE factorisation -- and must reproduce the oracle's analytic gradient (1e-7, the bar of the named
    kernels' own gradient test) and the named kernel's device gradient.  d = 40: both points from global memory."""
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d) * np.concatenate([np.linspace(0.8, 1.3, d), [1.1, 1.0]])
    ck = _lib.CustomKernel(SE_ARD_BODY, d + 1, fn=se_ard_fn)
    h = _lib.Handle(X, y, ck)
    h.set_option("profile", 1)
    h.reset_profile()
    ll, g, info = h.loglik_grad(th)
    one = h.profile()["kbuild"]["bytes"]
    assert info == 0 and h.get_option("grad_analytic") == 1
    want = orc.log_likelihood_grad("se_ard", th, X, y)
    np.testing.assert_allclose(g, want, rtol=1e-7, atol=1e-7 * n)
    assert close(ll, orc.log_likelihood("se_ard", th, X, y), n)
    if True:                                                  # (d = 40: the named kernel's windowed general reduction)
        ref = _lib.Handle(X, y, "se_ard")
        l0, g0, _ = ref.loglik_grad(th)
        np.testing.assert_allclose(g, g0, rtol=1e-9, atol=1e-9 * n)
        assert np.array_equal(ref.loglik_grad(th)[1], g0)   # (fixed summation order: windowed general reduction at d = 40)
        ref.close()
    assert np.array_equal(h.loglik_grad(th)[1], g)          # .. and the run-time compiled dual-number reduction
    # the factor of theta stays resident, as after the named kernels' gradient
    mu, var = h.predict(syn.make_test_points(20, d))
    mo, so = orc.predict_internal(ck, th, X, y, syn.make_test_points(20, d))
    np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9)
    # the difference route: 2 p + 1 kernel builds and factorisations for the same answer to ~1e-6
    h.set_option("custom_grad", 0)
    h.reset_profile()
    ll2, g2, info2 = h.loglik_grad(th)
    many = h.profile()["kbuild"]["bytes"]
    assert info2 == 0 and h.get_option("grad_analytic") == 0
    assert many >= (2 * len(th) + 1) * one * 0.99
    np.testing.assert_allclose(g2, want, rtol=2e-5, atol=2e-5 * n)
    h.close()


@pytest.mark.parametrize("body,fn,ncp,d,theta", [
    (NONSTAT_BODY, nonstat_fn, 3, 2, [0.9, 1.2, 0.7, 0.15]),
    (PERIODIC_BODY, periodic_fn, 3, 1, [1.3, 0.8, 1.1, 0.2]),
])
def test_gradient_of_nonstationary_and_cform_functions(body, fn, ncp, d, theta):
    n = 500
    X, y = syn.make_dataset(n, d)
    ck = _lib.CustomKernel(body, ncp, fn=fn)
    for mean in ("zero", "const"):
        th = np.array(theta + ([0.3] if mean == "const" else []))
        h = _lib.Handle(X, y, ck, mean=mean)
        ll, g, info = h.loglik_grad(th)
        assert info == 0 and h.get_option("grad_analytic") == 1
        np.testing.assert_allclose(g, _fd_grad(ck, th, X, y, mean), rtol=2e-6, atol=2e-6 * n)
        h.close()


def test_gradient_fp32_and_fallback_for_a_body_that_cannot_be_differentiated():
    n, d = 600, 3
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d, dtype="f32")
    ck = _lib.CustomKernel(SE_ARD_BODY, d + 1, fn=se_ard_fn)
    want = orc.log_likelihood_grad("se_ard", th, X, y)
    h32 = _lib.Handle(X, y, ck, dtype=32)
    ll, g, info = h32.loglik_grad(th)
    assert info == 0 and h32.get_option("grad_analytic") == 1
    np.testing.assert_allclose(g, want, rtol=2e-2, atol=2e-2 * np.abs(want).max())
    h32.set_option("custom_grad", 0)                          # fp32 differences: step 5e-3 |theta|, a few per cent
    ll, g, info = h32.loglik_grad(th)
    assert info == 0 and h32.get_option("grad_analytic") == 0
    np.testing.assert_allclose(g, want, rtol=0.1, atol=0.1 * np.abs(want).max())
    h32.close()
    # intermediates of a fixed scalar type: the value program compiles, the dual-number program does not -> differences
    fixed = ("double s = 0; for (int k = 0; k < D; ++k) { const double u = (double)(X(k) - Y(k)) / (double)P(k); s += u * u; } "
             "return (T)((double)P(D) * (double)P(D) * exp(-0.5 * s));")
    h = _lib.Handle(X, y, _lib.CustomKernel(fixed, d + 1, fn=se_ard_fn))
    th64 = syn.default_theta("se_ard", d)
    ll, g, info = h.loglik_grad(th64)
    assert info == 0 and h.get_option("grad_analytic") == 0
    np.testing.assert_allclose(g, orc.log_likelihood_grad("se_ard", th64, X, y), rtol=2e-5, atol=2e-5 * n)
    h.close()


def test_compile_call_and_create_share_the_code_object_cache():
    """gphip_custom_compile_d(.., d, ..) compiles THE program a d-dimensional handle runs (specialised on d), so a
    create after it -- or it after a create -- is a cache hit; the d-generic gphip_custom_compile is another code object.  Also
    the diagonal of a kernel written with Power(r2, 0.5) must not poison the dual-number gradient (pow at a zero base)."""
    import ctypes as C
    lib = _lib.load()
    d = 3
    body = SE_ARD_BODY + " /* cache test " + str(np.random.default_rng().integers(1 << 60)) + " */"
    hit = C.c_int(-1)
    assert lib.gphip_custom_compile_d(body.encode(), 64, None, -1, d, C.byref(hit)) == 0 and hit.value == 0
    X, y = syn.make_dataset(300, d)
    h = _lib.Handle(X, y, _lib.CustomKernel(body, d + 1, fn=se_ard_fn))
    assert lib.gphip_custom_compile_d(body.encode(), 64, None, -1, d, C.byref(hit)) == 0 and hit.value == 1
    assert lib.gphip_custom_compile(body.encode(), 64, None, -1, C.byref(hit)) == 0 and hit.value == 0     # d-generic: its own entry
    body2 = body + " /* second */"
    h2 = _lib.Handle(X, y, _lib.CustomKernel(body2, d + 1, fn=se_ard_fn))
    assert lib.gphip_custom_compile_d(body2.encode(), 64, None, -1, d, C.byref(hit)) == 0 and hit.value == 1   # the create compiled it
    h.close(); h2.close()
    # Matern-1/2 through Power(r2, 0.5): value and gradient finite and equal to central differences
    mbody = "T s = 0; for (int k = 0; k < D; ++k) { const T u = (X(k) - Y(k)) / P(0); s += u * u; } return Power(P(1),2) * Exp(-Power(s, 0.5));"
    ck = _lib.CustomKernel(mbody, 2, fn=lambda A, B, p: p[1] ** 2 * np.exp(-np.sqrt((((A - B) / p[0]) ** 2).sum(-1))))
    h = _lib.Handle(X, y, ck)
    th = np.array([0.8, 1.1, 0.2])
    ll, g, info = h.loglik_grad(th)
    assert info == 0 and np.all(np.isfinite(g)), g
    for k in range(3):
        e = np.zeros(3); e[k] = 1e-5 * th[k]
        num = (h.loglik(th + e)[0] - h.loglik(th - e)[0]) / (2 * e[k])
        assert abs(g[k] - num) <= 1e-5 * max(1.0, abs(num)), (k, g[k], num)
    assert h.get_option("grad_analytic") == 1                                     # (dual numbers, not the central-difference fallback)
    h.close()

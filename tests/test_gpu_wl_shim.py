"""Every LibraryLink entry point of csrc/librarylink_shim.cpp, compiled against the tests-only stand-in header and
driven on the GPU through a fake WolframLibraryData (tests/wl_stub/shim_driver.cpp) exactly as the Wolfram kernel
would call it: "Constant" argument tensors, MArgument arrays, results created by the library and handed over.
Checked: {value, info} packing, NaN rows for bad samples, vector/matrix "Inverse", the status -> LIBRARY_* map,
ownership (no result leaked on error paths, no "Constant" argument freed, strings disowned), dtype and device-list
pass-through -- and the numbers against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

from bayesianinference_amd import _lib, build, synthetic as syn
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

MT_INT, MT_REAL = 2, 3
NO_ERROR, TYPE_ERROR, RANK_ERROR, DIMENSION_ERROR, FUNCTION_ERROR = 0, 1, 2, 3, 6


class Kernel:
    """The calling side of LibraryLink: builds MArguments, owns argument tensors, receives results."""

    def __init__(self):
        _lib.load()
        self.lib = C.CDLL(build.build_wl_stub())
        L = self.lib
        L.drv_libdata.restype = C.c_void_p
        L.drv_tensor.restype = C.c_void_p
        L.drv_tensor.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
        L.drv_release.argtypes = [C.c_void_p]
        for f in ("drv_rank", "drv_type"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p]
        L.drv_dims.restype = C.POINTER(C.c_int64)
        L.drv_dims.argtypes = [C.c_void_p]
        L.drv_data.restype = C.c_void_p
        L.drv_data.argtypes = [C.c_void_p]
        for f in ("drv_live", "drv_disowned", "drv_const_frees"):
            getattr(L, f).restype = C.c_long
        self.data = C.c_void_p(L.drv_libdata())
        assert L.WolframLibrary_initialize(self.data) == 0

    def tensor(self, arr):
        arr = np.ascontiguousarray(arr)
        typ = MT_INT if arr.dtype.kind in "iu" else MT_REAL
        arr = arr.astype(np.int64 if typ == MT_INT else np.float64)
        dims = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
        return self.lib.drv_tensor(typ, arr.ndim, dims, arr.ctypes.data)

    def call(self, name, args, result="tensor"):
        """args: python ints (Integer), floats (Real), str (UTF8String), numpy arrays (tensors).
        Returns (return code, result value or None)."""
        fn = getattr(self.lib, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        cells, tensors = [], []
        for a in args:
            if isinstance(a, str):
                buf = C.create_string_buffer(a.encode())
                cells.append((C.c_char_p(C.addressof(buf)), buf))
            elif isinstance(a, (int, np.integer)):
                cells.append((C.c_int64(int(a)), None))
            elif isinstance(a, float):
                cells.append((C.c_double(a), None))
            else:
                t = self.tensor(a)
                tensors.append(t)
                cells.append((C.c_void_p(t), None))
        margs = (C.c_void_p * max(len(cells), 1))(*[C.addressof(c[0]) for c in cells])
        res_cell = {"tensor": C.c_void_p(0), "int": C.c_int64(-12345), "real": C.c_double(float("nan"))}[result]
        rc = fn(self.data, len(cells), margs, C.c_void_p(C.addressof(res_cell)))
        out = None
        if rc == NO_ERROR:
            if result == "tensor":
                t = res_cell.value
                rank = self.lib.drv_rank(t)
                shape = tuple(self.lib.drv_dims(t)[i] for i in range(rank))
                n = int(np.prod(shape)) if rank else 1
                typ = np.int64 if self.lib.drv_type(t) == MT_INT else np.float64
                out = np.ctypeslib.as_array(C.cast(self.lib.drv_data(t), C.POINTER(C.c_double if typ == np.float64 else C.c_int64)),
                                            shape=(n,)).reshape(shape).copy()
                self.lib.drv_release(t)                     # the kernel owned the result and drops it
            else:
                out = res_cell.value
        for t in tensors:
            self.lib.drv_release(t)
        return rc, out


def test_every_shim_entry_point_through_a_fake_wolfram_library_data():
    K = Kernel()
    n, d = 300, 3
    X, y = syn.make_dataset(n, d)
    th = syn.default_theta("se_ard", d)
    live0 = K.lib.drv_live()

    # create: dtype and device list pass through; bad shapes map to LIBRARY_* codes
    rc, h = K.call("gphip_wl_create", [X, y, 1, 0, 64, np.array([0])], "int")
    assert rc == NO_ERROR and h == 0
    rc, h32 = K.call("gphip_wl_create", [X, y, 3, 0, 32, np.zeros(0, np.int64)], "int")      # Matern-5/2 ARD, fp32, current device
    assert rc == NO_ERROR and h32 == 1
    rc, hg = K.call("gphip_wl_create", [X, y, 1, 0, 64, np.array([0, 0])], "int")            # two virtual ranks: multi-device handle
    assert rc == NO_ERROR and hg == 2
    assert K.call("gphip_wl_create", [X, y[:-1], 1, 0, 64, np.array([0])], "int")[0] == DIMENSION_ERROR
    assert K.call("gphip_wl_create", [X, X, 1, 0, 64, np.array([0])], "int")[0] == RANK_ERROR
    assert K.call("gphip_wl_create", [X, y, 99, 0, 64, np.array([0])], "int")[0] == TYPE_ERROR        # GPHIP_ERR_ARG
    assert K.call("gphip_wl_create", [X, y, 1, 0, 16, np.array([0])], "int")[0] == FUNCTION_ERROR    # unsupported dtype

    # options (UTF8String argument, disowned after use)
    assert K.call("gphip_wl_set_option", [h, "panel", 2.0], "int") == (NO_ERROR, 0)
    assert K.call("gphip_wl_set_option", [h, "no_such_option", 1.0], "int")[0] == TYPE_ERROR
    assert K.call("gphip_wl_set_option", [hg, "shard_min_n", 0.0], "int") == (NO_ERROR, 0)
    assert K.lib.drv_disowned() == 3

    # loglik: {value, info}; singular K is a RESULT (info = 1), never an error code
    want = orc.log_likelihood("se_ard", th, X, y)
    rc, r = K.call("gphip_wl_loglik", [h, th])
    assert rc == NO_ERROR and r.shape == (2,) and r[1] == 0 and abs(r[0] - want) <= 1e-8 * max(abs(want), n)
    rc, r = K.call("gphip_wl_loglik", [hg, th])                                           # sharded over the two ranks
    assert rc == NO_ERROR and r[1] == 0 and abs(r[0] - want) <= 1e-8 * max(abs(want), n)
    rc, r = K.call("gphip_wl_loglik", [h, np.array([np.nan, 1, 1, 1, 1.0])])
    assert rc == NO_ERROR and r.tolist() == [0.0, 2.0]
    assert K.call("gphip_wl_loglik", [h, th[:2]])[0] == DIMENSION_ERROR                   # wrong p
    assert K.call("gphip_wl_loglik", [77, th])[0] == FUNCTION_ERROR                       # unknown handle
    assert K.call("gphip_wl_loglik", [h, np.stack([th, th])])[0] == RANK_ERROR
    w32 = orc.log_likelihood("matern52_ard", th, X, y)
    rc, r = K.call("gphip_wl_loglik", [h32, th])
    assert rc == NO_ERROR and r[1] == 0 and abs(r[0] - w32) <= 1e-3 * max(abs(w32), n)    # fp32 device arithmetic

    Th = np.stack([th, th * 1.1, np.array([0.0, 1, 1, 1, 1.0]), th * 0.9])
    rc, r = K.call("gphip_wl_loglik_batch", [h, Th])
    assert rc == NO_ERROR and r.shape == (4, 2) and r[:, 1].tolist() == [0, 0, 2, 0] and r[2, 0] == 0.0
    for i in (0, 1, 3):
        w = orc.log_likelihood("se_ard", Th[i], X, y)
        assert abs(r[i, 0] - w) <= 1e-8 * max(abs(w), n)
    assert K.call("gphip_wl_loglik_batch", [h, th])[0] == RANK_ERROR

    rc, r = K.call("gphip_wl_loglik_grad", [h, th])
    g = orc.log_likelihood_grad("se_ard", th, X, y)
    assert rc == NO_ERROR and r.shape == (2 + len(th),) and r[1] == 0 and abs(r[0] - want) <= 1e-8 * max(abs(want), n)
    np.testing.assert_allclose(r[2:], g, rtol=1e-7, atol=1e-7 * np.abs(g).max())

    # fit -> info; then "Inverse" (vector AND matrix, BGP:194, 416), "LogDet", predict
    sing = th.copy(); sing[-1] = 0.0
    Xd = X.copy(); Xd[5] = Xd[200]
    rc, hd = K.call("gphip_wl_create", [Xd, y, 1, 0, 64, np.array([0])], "int")
    assert K.call("gphip_wl_logdet", [hd], "real")[0] == FUNCTION_ERROR                   # nothing fitted: GPHIP_ERR_STATE
    assert K.call("gphip_wl_fit", [hd, sing], "int") == (NO_ERROR, 1)                     # not SPD: a value
    assert K.call("gphip_wl_fit", [h, th], "int") == (NO_ERROR, 0)
    Kmat = orc.covariance_matrix("se_ard", th, X)
    rc, ld = K.call("gphip_wl_logdet", [h], "real")
    assert rc == NO_ERROR and abs(ld - np.linalg.slogdet(Kmat)[1]) <= 1e-8 * n
    rc, a = K.call("gphip_wl_solve", [h, y])
    assert rc == NO_ERROR and a.shape == (n,)
    np.testing.assert_allclose(a, np.linalg.solve(Kmat, y), rtol=1e-8, atol=1e-9)
    B = np.random.default_rng(1).standard_normal((n, 5))
    rc, a = K.call("gphip_wl_solve", [h, B])
    assert rc == NO_ERROR and a.shape == (n, 5)
    np.testing.assert_allclose(a, np.linalg.solve(Kmat, B), rtol=1e-8, atol=1e-9)
    assert K.call("gphip_wl_solve", [h, y[:10]])[0] == DIMENSION_ERROR

    Xs = syn.make_test_points(17, d)
    mo, so = orc.predict_internal("se_ard", th, X, y, Xs)
    rc, r = K.call("gphip_wl_predict", [h, Xs])
    assert rc == NO_ERROR and r.shape == (2, 17)
    np.testing.assert_allclose(r[0], mo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.sqrt(r[1]), so, rtol=1e-7)
    assert K.call("gphip_wl_predict", [hd, Xs])[0] == FUNCTION_ERROR                      # its fit failed

    rc, r = K.call("gphip_wl_predict_samples", [h, Th, Xs])
    assert rc == NO_ERROR and r.shape == (2, 4, 17)
    assert np.all(np.isnan(r[:, 2])) and np.all(np.isfinite(np.delete(r, 2, axis=1)))     # NaN rows for the bad sample
    m1, s1 = orc.predict_internal("se_ard", Th[1], X, y, Xs)
    np.testing.assert_allclose(r[0, 1], m1, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.sqrt(r[1, 1]), s1, rtol=1e-7)

    rc, r = K.call("gphip_wl_covariance", [h, th])
    assert rc == NO_ERROR and r.shape == (n, n)
    np.testing.assert_allclose(r, Kmat, rtol=1e-12)
    rc, r = K.call("gphip_wl_cross_covariance", [h, th, Xs])
    ko, kap = orc.k_and_kappa("se_ard", th, X, Xs)
    assert rc == NO_ERROR and r.shape == (n + 1, 17)
    np.testing.assert_allclose(r[:n], ko, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(r[n], kap, rtol=1e-15)

    for hh in (h, h32, hg, hd):
        assert K.call("gphip_wl_destroy", [hh], "int") == (NO_ERROR, 0)
    assert K.call("gphip_wl_loglik", [h, th])[0] == FUNCTION_ERROR                        # destroyed handle
    # ownership: every library-created tensor was either handed over (and released by the "kernel") or freed on
    # its error path; no "Constant" argument was freed
    assert K.lib.drv_live() == live0 and K.lib.drv_const_frees() == 0
    K.lib.WolframLibrary_uninitialize(K.data)


def test_new_shim_entry_points_pointwise_kernels_device_count_and_native_sampler():
    K = Kernel()
    n, d = 200, 2
    X, y = syn.make_dataset(n, d)
    live0 = K.lib.drv_live()
    rc, ndev = K.call("gphip_wl_device_count", [], "int")
    assert rc == NO_ERROR and ndev >= 1
    # a composed kernel through the integer code of GPHIP_KERNEL_COMPOSE: SE-ARD + Matern-3/2 + const
    kid = _lib.kernel_id("se_ard+matern32+const")
    rc, h = K.call("gphip_wl_create", [X, y, kid, 0, 64, np.array([0])], "int")
    assert rc == NO_ERROR
    th = np.array([0.6, 1.4, 0.8, 1.9, 0.7, 0.05, 0.2])
    rc, r = K.call("gphip_wl_loglik", [h, th])
    want = orc.log_likelihood("se_ard+matern32+const", th, X, y)
    assert rc == NO_ERROR and r[1] == 0 and abs(r[0] - want) <= 1e-8 * max(abs(want), n)

    # test points of the wrong width: LIBRARY_DIMENSION_ERROR, not an out-of-bounds read
    Xs = syn.make_test_points(9, d)
    assert K.call("gphip_wl_fit", [h, th], "int") == (NO_ERROR, 0)
    bad = syn.make_test_points(9, d + 1)
    assert K.call("gphip_wl_predict", [h, bad])[0] == DIMENSION_ERROR
    assert K.call("gphip_wl_predict_samples", [h, th[None, :], bad])[0] == DIMENSION_ERROR
    assert K.call("gphip_wl_cross_covariance", [h, th, bad])[0] == DIMENSION_ERROR
    assert K.call("gphip_wl_predict", [h, Xs.astype(np.int64)])[0] == TYPE_ERROR

    # point-dependent nugget / mean: values per theta; an empty list = the constant form
    nf = lambda x: 0.04 * (1.0 + x[0] ** 2)                   # noqa: E731
    mf = lambda x: 0.1 - 0.3 * x[1]                           # noqa: E731
    nug = np.array([nf(x) for x in X])
    mean = np.array([mf(x) for x in X])
    Th = np.stack([th, th * 1.05])
    empty = np.zeros(0)
    rc, r = K.call("gphip_wl_loglik_batch_pw", [h, Th, np.stack([mean, mean]), np.stack([nug, nug])])
    w0 = orc.log_likelihood("se_ard+matern32+const", Th[0], X, y, nugget_fn=nf, mean_fn=mf)
    assert rc == NO_ERROR and r.shape == (2, 2) and r[0, 1] == 0 and abs(r[0, 0] - w0) <= 1e-8 * max(abs(w0), n)
    rc, r = K.call("gphip_wl_loglik_batch_pw", [h, Th, empty, empty])               # both constant: the plain closure
    assert rc == NO_ERROR and abs(r[0, 0] - want) <= 1e-8 * max(abs(want), n)
    assert K.call("gphip_wl_loglik_batch_pw", [h, Th, mean, empty])[0] == DIMENSION_ERROR     # B x N expected
    assert K.call("gphip_wl_fit_pw", [h, th, mean, nug], "int") == (NO_ERROR, 0)
    Kmat = orc.covariance_matrix("se_ard+matern32+const", th, X, nugget_fn=nf)
    rc, ld = K.call("gphip_wl_logdet", [h], "real")
    assert rc == NO_ERROR and abs(ld - np.linalg.slogdet(Kmat)[1]) <= 1e-8 * n
    nug_s, mean_s = np.array([nf(x) for x in Xs]), np.array([mf(x) for x in Xs])
    rc, r = K.call("gphip_wl_predict_samples_pw", [h, th[None, :], mean[None, :], nug[None, :], Xs, mean_s[None, :], nug_s[None, :]])
    mo, so = orc.predict_internal("se_ard+matern32+const", th, X, y, Xs, nugget_fn=nf, mean_fn=mf)
    assert rc == NO_ERROR and r.shape == (2, 1, 9)
    np.testing.assert_allclose(r[0, 0], mo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.sqrt(r[1, 0]), so, rtol=1e-7)

    # the native sampler: rows {point.., loglik, logprior, acceptance rate}; the pool first (acceptance rate NaN)
    rc, hs = K.call("gphip_wl_create", [X[:64, :1].copy(), y[:64].copy(), 0, 0, 64, np.array([0])], "int")
    box = np.array([[0.05, 1.5], [0.2, 3.0], [0.03, 0.6]])
    opts = np.array([30, 120, 20, 10, 8, 0.01, 0.0, 1.0, 3.0])
    rc, rows = K.call("gphip_wl_nested_sampling", [hs, box, np.zeros(3, np.int64), opts, empty])
    assert rc == NO_ERROR and rows.ndim == 2 and rows.shape[1] == 6 and rows.shape[0] > 30
    assert np.all(np.isnan(rows[:30, 5])) and np.all(np.isfinite(rows[30:, 5]))
    assert np.all((rows[:, :3] >= box[:, 0]) & (rows[:, :3] <= box[:, 1]))
    ll0, info = _lib.Handle(X[:64, :1], y[:64], "se").loglik(rows[40, :3])
    assert info == 0 and abs(ll0 - rows[40, 3]) <= 1e-8 * max(abs(ll0), 64)
    np.testing.assert_allclose(rows[:, 4], -np.sum(np.log(box[:, 1] - box[:, 0])))            # uniform prior density
    start = rows[:30, :3].copy()
    rc, rows2 = K.call("gphip_wl_nested_sampling", [hs, box, np.zeros(3, np.int64), opts, start])
    assert rc == NO_ERROR and np.array_equal(rows2[:30, :3], start)
    assert K.call("gphip_wl_nested_sampling", [hs, box[:2], np.zeros(3, np.int64), opts, empty])[0] == DIMENSION_ERROR
    assert K.call("gphip_wl_nested_sampling", [hs, box, np.zeros(3, np.int64), opts[:5], empty])[0] == DIMENSION_ERROR

    for hh in (h, hs):
        assert K.call("gphip_wl_destroy", [hh], "int") == (NO_ERROR, 0)
    assert K.lib.drv_live() == live0 and K.lib.drv_const_frees() == 0
    K.lib.WolframLibrary_uninitialize(K.data)


def test_native_sampler_with_a_tabulated_normal_prior_matches_quadrature_over_30_seeds():
    """nestedSamplingHIP for a NON-uniform (separable) prior: gphip_wl_nested_sampling_tab takes each factor's log density as a
    table over the parameter's range (what GPHIP.wl sends for a ProductDistribution of univariate distributions) and a
    starting pool drawn from the prior.  Null kernel + constant mean: log L(sn, mu) is analytic, so the evidence under
    sn ~ U[0.4, 2], mu ~ N(0.2, 0.5) truncated to [-1, 1.5] is a smooth 2-D integral (midpoint rule, 1200 x 1200)."""
    import math
    from bayesianinference_amd import nested_sampling as ns
    K = Kernel()
    rng = np.random.default_rng(3)
    n = 40
    X = rng.random((n, 1))
    y = 0.3 + 0.8 * rng.standard_normal(n)
    box = np.array([[0.4, 2.0], [-1.0, 1.5]])
    m0, s0 = 0.2, 0.5
    from scipy.stats import norm
    mass = norm.cdf(box[1, 1], m0, s0) - norm.cdf(box[1, 0], m0, s0)

    def logprior(sn, mu):                                    # the density the TABLES describe (normalised on the box)
        return -math.log(box[0, 1] - box[0, 0]) + norm.logpdf(mu, m0, s0) - math.log(mass)

    g = 1200
    sn = box[0, 0] + (np.arange(g) + 0.5) * (box[0, 1] - box[0, 0]) / g
    mu = box[1, 0] + (np.arange(g) + 0.5) * (box[1, 1] - box[1, 0]) / g
    s1, s2 = y.sum(), (y * y).sum()
    quad = (s2 - 2 * mu[None, :] * s1 + n * mu[None, :] ** 2) / sn[:, None] ** 2
    ll = -0.5 * (n * math.log(2 * math.pi) + 2 * n * np.log(sn)[:, None] + quad)
    lp = -math.log(box[0, 1] - box[0, 0]) + norm.logpdf(mu, m0, s0)[None, :] - math.log(mass)
    cell = (box[0, 1] - box[0, 0]) / g * (box[1, 1] - box[1, 0]) / g
    want = ns.log_sum_exp((ll + lp).ravel()) + math.log(cell)
    nodes = 513
    tab = np.stack([np.full(nodes, -math.log(box[0, 1] - box[0, 0])),
                    norm.logpdf(np.linspace(box[1, 0], box[1, 1], nodes), m0, s0) - math.log(mass)])
    rc, hs = K.call("gphip_wl_create", [X, y, 4, 1, 64, np.array([0])], "int")          # null kernel, constant mean
    assert rc == NO_ERROR
    pool, zs = 60, []
    for seed in range(30):
        r = np.random.default_rng(100 + seed)
        start = np.empty((pool, 2))
        start[:, 0] = r.uniform(box[0, 0], box[0, 1], pool)
        k = 0
        while k < pool:                                       # truncated normal by rejection (RandomVariate + Select in GPHIP.wl)
            v = r.normal(m0, s0)
            if box[1, 0] <= v <= box[1, 1]:
                start[k, 1] = v
                k += 1
        opts = np.array([pool, 10000, 100, 25, 32, 0.01, 0.0, 1.0, float(seed)])
        # odd seeds: no starting points -- the library draws the pool from the tables itself (what GPHIP.wl does by default)
        given = seed % 2 == 0
        rc, rows = K.call("gphip_wl_nested_sampling_tab", [hs, box, tab, opts, start if given else np.zeros(0)])
        assert rc == NO_ERROR and rows.shape[1] == 5 and rows.shape[0] > pool
        if given:
            np.testing.assert_array_equal(rows[:pool, :2], start)
        else:
            assert np.all((rows[:pool, :2] >= box[:, 0]) & (rows[:pool, :2] <= box[:, 1]))
            assert abs(rows[:pool, 1].mean() - m0) < 0.35             # (drawn from N(0.2, 0.5) cut to the box, not uniformly)
        i = pool + 7                                          # the recorded prior density IS the tabulated one (interpolated)
        assert abs(rows[i, 3] - logprior(rows[i, 0], rows[i, 1])) < 1e-8
        res = {"Points": rows[:, :2], "LogLikelihood": rows[:, 2], "LogPriorPDF": rows[:, 3], "AcceptanceRate": rows[:, 4],
               "SamplePoolSize": pool, "GeneratedNestedSamples": len(rows) - pool, "TotalSamples": len(rows)}
        out = ns.evidence_sampling(res, ["sn", "mu"], pool, np.random.default_rng(seed))
        zs.append((out["LogEvidence"]["Mean"] - want) / out["LogEvidence"]["StandardError"])
    zs = np.array(zs)
    # (30 seeds: the mean of unbiased z-scores has sigma 0.18; 60-seed runs of scripts/gpu_tab_pool_bias.py: -0.02 +- 0.14 with
    #  the pool given, -0.20 +- 0.13 with the pool drawn by the library)
    assert abs(zs.mean()) < 0.55 and np.all(np.abs(zs) < 4.0), zs
    assert 0.5 < zs.std(ddof=1) < 2.0, zs
    # argument checks: table row count, too few nodes, missing pool
    assert K.call("gphip_wl_nested_sampling_tab", [hs, box, tab[:1], opts, start])[0] == DIMENSION_ERROR
    assert K.call("gphip_wl_nested_sampling_tab", [hs, box, tab[:, :3], opts, start])[0] == DIMENSION_ERROR
    assert K.call("gphip_wl_nested_sampling_tab", [hs, box, tab, opts, start[:, :1]])[0] == DIMENSION_ERROR
    assert K.call("gphip_wl_destroy", [hs], "int") == (NO_ERROR, 0)


def test_create_custom_through_the_shim_the_way_gphip_wl_drives_it():
    """gphip_wl_create_custom with a CForm-style body, then exactly GPHIP.wl's calling pattern for an arbitrary kernel: theta is
    the reference's parameter vector plus a dummy sigma_n slot, nugget and mean arrive as VALUES per point (the reference's
    expressions evaluated on the host)."""
    K = Kernel()
    n, d = 220, 2
    X, y = syn.make_dataset(n, d)
    live0, dis0 = K.lib.drv_live(), K.lib.drv_disowned()
    # Function[{p, q}, sf^2 Exp[-(p - q).(p - q)/(2 l^2)] (1 + c^2 p[[1]] q[[1]])] printed by CForm over {l, sf, c, sn}
    # -- as ToString[CForm[..]] prints it with the package's stand-in symbols (their context is not on $ContextPath)
    cform = ("(Power(GPHIP_Private_gphipPc1,2)*(1 + Power(GPHIP_Private_gphipPc2,2)*GPHIP_Private_gphipXc0*GPHIP_Private_gphipYc0))/"
             "Power(E,(Power(GPHIP_Private_gphipXc0 - GPHIP_Private_gphipYc0,2) + Power(GPHIP_Private_gphipXc1 - GPHIP_Private_gphipYc1,2))/"
             "(2.*Power(GPHIP_Private_gphipPc0,2)))")
    body = ("return (Power(P(1),2)*(1 + Power(P(2),2)*X(0)*Y(0)))/"
            "Power(E,(Power(X(0) - Y(0),2) + Power(X(1) - Y(1),2))/(2.*Power(P(0),2)));")        # what the shim makes of it
    fn = lambda A, B, p: p[1] ** 2 * np.exp(-0.5 * ((A - B) ** 2).sum(-1) / p[0] ** 2) * (1.0 + p[2] ** 2 * A[..., 0] * B[..., 0])  # noqa: E731
    ck = _lib.CustomKernel(body, 4, fn=fn)                     # all FOUR reference parameters are P(k); sn = P(3) is used by the nugget only
    rc, h = K.call("gphip_wl_create_custom", [X, y, cform, 4, 0, 64, np.array([0])], "int")
    assert rc == NO_ERROR and K.lib.drv_disowned() == dis0 + 1
    theta = np.array([0.9, 1.2, 0.7, 0.15])
    lifted = np.append(theta, 1.0)[None, :]                    # GPHIP.wl: Join[theta, {1.}]
    nug = np.full((1, n), theta[3] ** 2)
    rc, r = K.call("gphip_wl_loglik_batch_pw", [h, lifted, np.zeros(0), nug])
    want = orc.log_likelihood(ck, np.append(theta, theta[3]), X, y)
    assert rc == NO_ERROR and r[0, 1] == 0 and abs(r[0, 0] - want) <= 1e-8 * max(abs(want), n)
    Xs = syn.make_test_points(7, d)
    rc, r = K.call("gphip_wl_predict_samples_pw", [h, lifted, np.zeros(0), nug, Xs, np.zeros(0), np.full((1, 7), theta[3] ** 2)])
    mo, so = orc.predict_internal(ck, np.append(theta, theta[3]), X, y, Xs)
    assert rc == NO_ERROR
    np.testing.assert_allclose(r[0, 0], mo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.sqrt(r[1, 0]), so, rtol=1e-7)
    assert K.call("gphip_wl_destroy", [h], "int") == (NO_ERROR, 0)
    # a body that does not compile: an error code (GPHIP.wl then falls back to the reference's own path), nothing leaked
    assert K.call("gphip_wl_create_custom", [X, y, "Undefined(gphipPc0)", 1, 0, 64, np.array([0])], "int")[0] == FUNCTION_ERROR
    assert K.call("gphip_wl_create_custom", [X, y, 'Foo("x") + gphipPc0', 1, 0, 64, np.array([0])], "int")[0] == FUNCTION_ERROR   # no C expression
    assert K.call("gphip_wl_create_custom", [X, y[:-1], cform, 4, 0, 64, np.array([0])], "int")[0] == DIMENSION_ERROR
    # the kernel-name grammar the package asks the library for
    rc, spec = K.call("gphip_wl_kernel_spec", ["SEARD + Matern32 + Const", 3])
    assert rc == NO_ERROR and list(spec) == [_lib.kernel_id("se_ard+matern32+const"), 1, 1, 5, 1, 4, 2, 7]
    rc, spec = K.call("gphip_wl_kernel_spec", ["NotAKernel", 3])
    assert rc == NO_ERROR and list(spec) == [-1] * 8
    assert K.lib.drv_live() == live0 and K.lib.drv_const_frees() == 0
    K.lib.WolframLibrary_uninitialize(K.data)


def test_native_sampler_with_a_joint_prior_through_a_library_callback():
    """nestedSamplingHIP for a JOINT (non-separable) prior: the reference's "LogPriorPDFFunction" -- a CompiledFunction of one real
    vector (BS:412-427) -- is connected with ConnectLibraryCallbackFunction["gphip_logprior", f] and evaluated by the native driver
    through callLibraryCallbackFunction (gphip_wl_nested_sampling_cb).  Here the fake kernel connects a C function pointer standing
    in for the CompiledFunction: a CORRELATED bivariate normal over (sn, mu), cut to the box.  Null kernel + constant mean: the
    evidence is a smooth 2-D integral (midpoint rule, 1200 x 1200); z-scores over 24 seeds."""
    import math
    from bayesianinference_amd import nested_sampling as ns
    K = Kernel()
    L = K.lib
    CFN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_int64)
    L.drv_connect_callback.argtypes = [C.c_char_p, CFN, C.c_int]
    L.drv_cb_released.restype = L.drv_cb_calls.restype = C.c_long
    rng = np.random.default_rng(5)
    n = 40
    X = rng.random((n, 1))
    y = 0.3 + 0.8 * rng.standard_normal(n)
    box = np.array([[0.4, 2.0], [-1.0, 1.5]])
    m = np.array([1.0, 0.3])
    S = np.array([[0.16, 0.09], [0.09, 0.25]])               # correlation 0.45: not a product of its marginals
    Si = np.linalg.inv(S)
    g = 1200
    sn = box[0, 0] + (np.arange(g) + 0.5) * (box[0, 1] - box[0, 0]) / g
    mu = box[1, 0] + (np.arange(g) + 0.5) * (box[1, 1] - box[1, 0]) / g
    dsn, dmu = sn[:, None] - m[0], mu[None, :] - m[1]
    lq = -0.5 * (Si[0, 0] * dsn * dsn + 2 * Si[0, 1] * dsn * dmu + Si[1, 1] * dmu * dmu)
    cell = (box[0, 1] - box[0, 0]) / g * (box[1, 1] - box[1, 0]) / g
    lognorm = ns.log_sum_exp(lq.ravel()) + math.log(cell)     # the prior is the Gaussian restricted to the box, normalised there
    s1, s2 = y.sum(), (y * y).sum()
    quad = (s2 - 2 * mu[None, :] * s1 + n * mu[None, :] ** 2) / sn[:, None] ** 2
    ll = -0.5 * (n * math.log(2 * math.pi) + 2 * n * np.log(sn)[:, None] + quad)
    want = ns.log_sum_exp((ll + lq - lognorm).ravel()) + math.log(cell)
    calls = [0]

    def logprior(th, p):
        calls[0] += 1
        if p != 2 or not (box[0, 0] <= th[0] <= box[0, 1] and box[1, 0] <= th[1] <= box[1, 1]):
            return -1.7976931348623157e308                    # $MachineLogZero outside the constraints
        v = np.array([th[0] - m[0], th[1] - m[1]])
        return float(-0.5 * v @ Si @ v - lognorm)
    cf = CFN(logprior)
    rc, hs = K.call("gphip_wl_create", [X, y, 4, 1, 64, np.array([0])], "int")
    assert rc == NO_ERROR
    opts0 = np.array([60, 10000, 100, 25, 32, 0.01, 0.0, 1.0, 0.0])
    start0 = np.tile(np.array([[1.0, 0.3]]), (60, 1))
    # no function connected yet: an error, not a crash
    assert K.call("gphip_wl_nested_sampling_cb", [hs, box, opts0, start0])[0] == FUNCTION_ERROR
    # a function of the wrong shape (rank-2 argument) is refused by the manager; the right one is accepted
    assert L.drv_connect_callback(b"gphip_logprior", cf, 1) == 0
    assert K.call("gphip_wl_nested_sampling_cb", [hs, box, opts0, start0])[0] == FUNCTION_ERROR
    assert L.drv_connect_callback(b"no_such_manager", cf, 0) == -1
    assert L.drv_connect_callback(b"gphip_logprior", cf, 0) == 1
    pool, zs = 60, []
    Lc = np.linalg.cholesky(S)
    for seed in range(24):
        r = np.random.default_rng(300 + seed)
        start = np.empty((pool, 2))
        k = 0
        while k < pool:                                       # RandomVariate[prior, ..] + Select inside the box
            v = m + Lc @ r.standard_normal(2)
            if box[0, 0] <= v[0] <= box[0, 1] and box[1, 0] <= v[1] <= box[1, 1]:
                start[k] = v
                k += 1
        opts = np.array([pool, 10000, 100, 25, 32, 0.01, 0.0, 1.0, float(seed)])
        rc, rows = K.call("gphip_wl_nested_sampling_cb", [hs, box, opts, start])
        assert rc == NO_ERROR and rows.shape[1] == 5 and rows.shape[0] > pool
        np.testing.assert_array_equal(rows[:pool, :2], start)
        i = pool + 5
        assert abs(rows[i, 3] - logprior(rows[i, :2], 2)) < 1e-12          # the recorded prior density is the callback's value
        res = {"Points": rows[:, :2], "LogLikelihood": rows[:, 2], "LogPriorPDF": rows[:, 3], "AcceptanceRate": rows[:, 4],
               "SamplePoolSize": pool, "GeneratedNestedSamples": len(rows) - pool, "TotalSamples": len(rows)}
        out = ns.evidence_sampling(res, ["sn", "mu"], pool, np.random.default_rng(seed))
        zs.append((out["LogEvidence"]["Mean"] - want) / out["LogEvidence"]["StandardError"])
    zs = np.array(zs)
    assert abs(zs.mean()) < 0.6 and np.all(np.abs(zs) < 4.0), zs
    assert 0.5 < zs.std(ddof=1) < 2.0, zs
    assert calls[0] > 24 * 1000 and L.drv_cb_calls() == calls[0] - 24     # (24 direct calls in the check above)
    # connecting another function releases the previous one; argument checks; uninitialize releases the last one
    rel0 = L.drv_cb_released()
    assert L.drv_connect_callback(b"gphip_logprior", cf, 0) == 1 and L.drv_cb_released() == rel0 + 1
    assert K.call("gphip_wl_nested_sampling_cb", [hs, box[:1], opts0, start0])[0] == DIMENSION_ERROR
    assert K.call("gphip_wl_nested_sampling_cb", [hs, box, opts0[:5], start0])[0] == DIMENSION_ERROR
    assert K.call("gphip_wl_nested_sampling_cb", [hs, box, opts0, start0[:, :1]])[0] == DIMENSION_ERROR
    assert K.call("gphip_wl_destroy", [hs], "int") == (NO_ERROR, 0)
    K.lib.WolframLibrary_uninitialize(K.data)
    assert L.drv_cb_released() == rel0 + 2

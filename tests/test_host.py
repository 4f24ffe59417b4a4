"""CPU tests of the host-side mirror (no GPU): object model, data normal form, priors, the
defineInferenceProblem smoke-test contract (BS:276-298) and sharding maps."""
import math

import numpy as np
import pytest
from scipy import stats

from bayesianinference_amd import distributed as D
from bayesianinference_amd import gaussian_process as gp


def test_inference_object_semantics():
    obj = gp.inferenceObject({"A": 1, "B": {"C": 2}})
    assert obj["A"] == 1 and obj["B", "C"] == 2
    assert obj["Properties"] == ["A", "B", "Properties"]
    assert gp.inferenceObject(obj).normal() == obj.normal()          # idempotent, BU:128
    assert obj.append({"Z": 3})["Z"] == 3 and "Z" not in obj
    failed = gp.inferenceObject(None)
    assert failed.failed and repr(failed) == "inferenceObject[$Failed]"


def test_data_normal_form():
    assert gp.dataNormalForm([1.0, 2.0, 3.0]).shape == (3, 1)         # BU:206
    X, Y = gp.dataNormalForm(([1.0, 2.0], [3.0, 4.0]))
    assert X.shape == (2, 1) and Y.shape == (2, 1)
    X, Y = gp.dataNormalForm([([0.0, 1.0], 2.0), ([1.0, 2.0], 3.0)])   # {x -> y ..}, BU:207
    assert X.shape == (2, 2) and Y.shape == (2, 1)
    assert gp.dataNormalForm(([1.0, 2.0], [3.0])) is None             # length mismatch -> $Failed
    assert gp.dataNormalForm("nonsense") is None


def test_kernel_names_and_wl_expressions():
    assert gp._resolve_kernel("SEARD") == "se_ard" and gp._resolve_kernel(None) == "null"
    assert gp._resolve_kernel("Matern-52") == "matern52"
    with pytest.raises(ValueError):
        gp._resolve_kernel(lambda p, q: 0.0)
    assert set(gp.WL_KERNEL_EXPRESSIONS) == {"se", "se_ard", "matern52", "matern52_ard", "null", "matern32", "matern32_ard",
                                             "rq", "rq_ard"}
    # composed forms: term [(+|*) term] [+const] (the reference's own example is a constant plus an SE, BGP:16)
    assert gp._resolve_kernel("SE + Const") == "se+const" and gp._resolve_kernel("SEARD*RQ") == "se_ard*rq"
    assert gp._resolve_kernel("Matern32ARD+SE+const") == "matern32_ard+se+const"
    assert gp.wl_kernel_expression("se+const").startswith("Function[{p, q}, c + (Function[{p, q}, sf^2 Exp[")
    for bad in ("se+null", "const", "se+", "se*foo"):
        with pytest.raises(ValueError):
            gp._resolve_kernel(bad)


def test_priors_and_random_domain_points():
    params = [("l", 0.1, 10.0), ("sf", 0.1, 10.0), ("sn", 0.01, 1.0)]
    lp = gp._log_prior_function("Uniform", params)
    assert lp([1.0, 1.0, 0.1]) == pytest.approx(-math.log(9.9 * 9.9 * 0.99))
    assert lp([20.0, 1.0, 0.1]) == gp.MACHINE_LOG_ZERO
    lp2 = gp._log_prior_function([stats.lognorm(1.0)] * 3, params)
    assert lp2([1.0, 1.0, 0.1]) == pytest.approx(sum(stats.lognorm(1.0).logpdf(t) for t in (1.0, 1.0, 0.1)))
    pts = gp.random_domain_points(params, 100)
    lo, hi = np.array([p[1] for p in params]), np.array([p[2] for p in params])
    assert pts.shape == (100, 3) and np.all(pts >= lo) and np.all(pts <= hi)


def test_define_inference_problem_smoke_contract():
    params = [("a", -1.0, 1.0), ("b", 0.0, 2.0)]
    good = gp.defineInferenceProblem({"Parameters": params, "PriorDistribution": "Uniform",
                                      "LogLikelihoodFunction": lambda t: -float(np.sum(np.square(t)))})
    assert not good.failed and good["ParameterSymbols"] == ["a", "b"]
    assert callable(good["LogPriorPDFFunction"])
    # a closure that is not total over the box fails the definition (BS:290-296)
    bad = gp.defineInferenceProblem({"Parameters": params, "PriorDistribution": "Uniform",
                                     "LogLikelihoodFunction": lambda t: float("nan")})
    assert bad.failed
    assert gp.defineInferenceProblem({"Parameters": params}).failed


def test_define_gaussian_process_argument_guards_need_no_gpu():
    X = np.zeros((4, 2))
    assert gp.defineGaussianProcess((X, np.zeros((4, 2))), "SEARD", variables=[("l", 0, 1)]).failed  # BGP:220
    assert gp.defineGaussianProcess((X, np.zeros(3)), "SEARD", variables=[("l", 0, 1)]).failed       # lengths
    assert gp.defineGaussianProcess("junk", "SEARD", variables=[("l", 0, 1)]).failed


def test_sharding_maps():
    assert list(D.shard_indices(7, 1, 3)) == [1, 4]
    allidx = np.sort(np.concatenate([D.shard_indices(10, r, 4) for r in range(4)]))
    assert list(allidx) == list(range(10))
    assert [D.block_cyclic_owner(j, 8) for j in (0, 7, 8, 9)] == [0, 7, 0, 1]
    assert list(D.local_block_columns(10, 2, 4)) == [2, 6]
    v, i = D.sharded_map(lambda th: (th.sum(axis=1), np.zeros(len(th), dtype=int)), np.ones((5, 3)))
    assert list(v) == [3.0] * 5 and not i.any()


def test_normalize_data_and_posterior_fraction():
    rng = np.random.default_rng(0)
    X = rng.normal(3.0, 2.0, (50, 2))
    Y = rng.normal(-1.0, 0.5, 50)
    nd = gp.normalizeData(X, Y)
    assert gp.normalizedDataQ(nd) and gp.normalizedDataQ(nd["Input"]) and not gp.normalizedDataQ({"a": 1})
    Z = nd["Input"]["NormalizedData"]
    np.testing.assert_allclose(Z.mean(axis=0), 0.0, atol=1e-12)
    np.testing.assert_allclose(Z.std(axis=0, ddof=1), 1.0, rtol=1e-12)          # StandardizedVector
    np.testing.assert_allclose(nd["Input"]["InverseFunction"](Z), X, rtol=1e-12)
    np.testing.assert_allclose(nd["Output"]["Function"](Y)[:, 0], nd["Output"]["NormalizedData"][:, 0])
    samples = [{"Point": [i], "CrudePosteriorWeight": w, "CrudeLogPosteriorWeight": math.log(w)}
               for i, w in enumerate([0.1, 0.5, 0.15, 0.25])]
    obj = gp.inferenceObject({"Samples": samples})
    top = gp.takePosteriorFraction(obj, 0.7)["Samples"]                          # BU:298-316
    assert [s["Point"][0] for s in top] == [1, 3]
    assert [s["Point"][0] for s in gp.takePosteriorFraction(obj, 1)["Samples"]] == [1, 3, 2, 0]


def test_mixture_percentiles_and_plot_moments():
    """What regressionPlot1D draws from the per-point MixtureDistribution (BV:303-374): InverseCDF at the
    percentile levels (default {0.95, 0.5, 0.05}) or the "Moments" triple with the real cube root of the
    third central moment.  Pure host arithmetic: checked against the CDF itself, closed forms and quadrature."""
    from scipy.special import ndtr, ndtri
    rng = np.random.default_rng(3)
    S, M = 9, 13
    pred = {"Weights": rng.random(S), "Mean": rng.normal(size=(S, M)), "StandardDeviation": 0.1 + rng.random((S, M))}
    levels = (0.95, 0.5, 0.05)
    q = gp.mixture_percentiles(pred, levels)
    w = pred["Weights"] / pred["Weights"].sum()
    for k, lev in enumerate(levels):
        cdf = np.einsum("s,sm->m", w, ndtr((q[k][None, :] - pred["Mean"]) / pred["StandardDeviation"]))
        np.testing.assert_allclose(cdf, lev, rtol=0, atol=1e-13)
    assert np.all(q[0] > q[1]) and np.all(q[1] > q[2])
    # one component: plain normal quantiles; zero-weight components are ignored
    one = {"Weights": np.array([2.0, 0.0]), "Mean": np.array([[1.5, -2.0], [9.0, 9.0]]),
           "StandardDeviation": np.array([[0.5, 3.0], [1.0, 1.0]])}
    np.testing.assert_allclose(gp.mixture_percentiles(one, (0.9, 0.2)),
                               np.array([[1.5, -2.0]]) + ndtri(np.array([[0.9], [0.2]])) * np.array([[0.5, 3.0]]), rtol=1e-12)
    with pytest.raises(ValueError):
        gp.mixture_percentiles(one, (0.5, 1.0))
    # "Moments": symmetric single normal -> {m + s, m, m - s}; skewed mixture -> third moment by quadrature
    np.testing.assert_allclose(gp.mixture_plot_moments(one), [[2.0, 1.0], [1.5, -2.0], [1.0, -5.0]], atol=1e-12)
    hi, mid, lo = gp.mixture_plot_moments(pred)
    m, v = gp.mixture_moments(pred)
    np.testing.assert_allclose(mid, m, rtol=1e-14)
    x = np.linspace(-12, 12, 200001)
    j = 4
    pdf = sum(w[s] * np.exp(-0.5 * ((x - pred["Mean"][s, j]) / pred["StandardDeviation"][s, j]) ** 2)
              / (pred["StandardDeviation"][s, j] * np.sqrt(2 * np.pi)) for s in range(S))
    m3 = np.trapezoid((x - m[j]) ** 3 * pdf, x)
    np.testing.assert_allclose(hi[j], m[j] + np.sqrt(v[j]) + np.cbrt(m3), rtol=1e-8)
    np.testing.assert_allclose(lo[j], m[j] - np.sqrt(v[j]) + np.cbrt(m3), rtol=1e-8)


def test_laplace_log_evidence_formula():
    """LA:22-30: max + (p log 2pi - log det P)/2, Missing[] unless det P > 0 -- exact for a Gaussian density."""
    from bayesianinference_amd import laplace
    P = np.array([[4.0, 1.0], [1.0, 3.0]])
    # unnormalised Gaussian exp(c - x'Px/2): Z = e^c (2pi)^(p/2) det(P)^(-1/2)
    assert laplace.laplaceLogEvidence(1.5, P) == pytest.approx(1.5 + np.log(2 * np.pi) - 0.5 * np.log(11.0), rel=1e-14)
    assert laplace.laplaceLogEvidence(0.0, 2.0) == pytest.approx(0.5 * (np.log(2 * np.pi) - np.log(2.0)))
    assert laplace.laplaceLogEvidence(0.0, [[1.0, 2.0], [2.0, 1.0]]) is None        # det < 0

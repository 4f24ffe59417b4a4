"""GPU test of the nested-sampling driver on the HIP likelihood: log evidence of a small GP against
brute-force midpoint integration of the same (batched) likelihood over the prior box."""
import math

import numpy as np
import pytest

from bayesianinference_amd import gaussian_process as gp, nested_sampling as ns, synthetic as syn

pytestmark = pytest.mark.gpu


def test_gp_evidence_matches_grid_integration():
    X, y = syn.make_dataset(96, 1)
    variables = [("l", 0.05, 1.5), ("sf", 0.2, 3.0), ("sn", 0.03, 0.6)]
    obj = gp.defineGaussianProcess((X, y), "SE", variables=variables, variablePrior="Uniform")
    assert not obj.failed
    ll = obj["LogLikelihoodFunction"]
    g = 36
    axes = [lo + (np.arange(g) + 0.5) * (hi - lo) / g for _, lo, hi in variables]
    grid = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, 3)
    vals = np.concatenate([ll(grid[i:i + 4096]) for i in range(0, len(grid), 4096)])
    want = ns.log_sum_exp(vals) - math.log(len(grid))          # uniform prior: Z = mean of L over the box
    res = ns.nestedSampling(obj, SamplePoolSize=100, MonteCarloSteps=30, Walkers=32, Seed=11)
    z, se = res["LogEvidence"]["Mean"], res["LogEvidence"]["StandardError"]
    assert abs(z - want) < 4 * se + 0.25, (z, se, want)
    # the sampled object drops straight into prediction (BGP:343-376 needs "Samples" + weights)
    pred = gp.predictFromGaussianProcess(ns.inferenceObject_take(res, 20), np.linspace(-1, 1, 5))
    assert pred["Mean"].shape == (20, 5) and np.all(np.isfinite(pred["Mean"]))


# ---------------------------------------------------------------------------------------------
# the NATIVE driver (gphip_nested_sampling): the same algorithm in C++ behind the C ABI, reachable from the WL package
# ---------------------------------------------------------------------------------------------
def _null_kernel_problem(n=40):
    """Null kernel (K = sn^2 I) with a constant mean: log L(sn, mu) = sum log N(y_i; mu, sn^2) is analytic, so the evidence
    under a uniform prior on the box is a smooth 2-D integral -- evaluated here on a 1500 x 1500 midpoint grid."""
    from bayesianinference_amd import _lib
    rng = np.random.default_rng(3)
    X = rng.random((n, 1))
    y = 0.3 + 0.8 * rng.standard_normal(n)
    box = np.array([[0.4, 2.0], [-1.0, 1.5]])
    g = 1500
    sn = box[0, 0] + (np.arange(g) + 0.5) * (box[0, 1] - box[0, 0]) / g
    mu = box[1, 0] + (np.arange(g) + 0.5) * (box[1, 1] - box[1, 0]) / g
    s1, s2 = y.sum(), (y * y).sum()
    quad = (s2 - 2 * mu[None, :] * s1 + n * mu[None, :] ** 2) / sn[:, None] ** 2
    ll = -0.5 * (n * math.log(2 * math.pi) + 2 * n * np.log(sn)[:, None] + quad)
    want = ns.log_sum_exp(ll.ravel()) - math.log(g * g)
    return _lib.Handle(X, y, "null", "const"), box, want


def test_native_sampler_log_evidence_is_unbiased_over_20_seeds():
    h, box, want = _null_kernel_problem()
    zs = []
    for seed in range(20):
        res = h.nested_sampling(box, pool=60, mc_steps=25, walkers=32, seed=seed)
        assert res["TotalSamples"] > 60 and np.all(np.isfinite(res["LogLikelihood"]))
        out = ns.evidence_sampling(res, ["sn", "mu"], 60, np.random.default_rng(seed))
        assert out["CrudeLogEvidence"] == pytest.approx(res["CrudeLogEvidence"], abs=1e-9)
        zs.append((out["LogEvidence"]["Mean"] - want) / out["LogEvidence"]["StandardError"])
    zs = np.array(zs)
    assert abs(zs.mean()) < 0.5, zs
    assert 0.5 < zs.std(ddof=1) < 2.0, zs
    h.close()


@pytest.mark.parametrize("walkers", [1, 32])
def test_native_sampler_prior_mass_shrinks_like_minus_i_over_n(walkers):
    """One data point y = 0, null kernel, sn pinned by a hair-thin box: log L = const - mu^2 / 2 under mu ~ U[-1, 1], so
    the prior mass above a dead point is |mu| exactly and log X_true(i) + i/n is a sum of i errors of sd 1/n."""
    from bayesianinference_amd import _lib
    h = _lib.Handle(np.zeros((1, 1)), np.zeros(1), "null", "const")
    box = np.array([[1.0, 1.00001], [-1.0, 1.0]])
    n = 50
    devs = []
    for seed in range(20):
        res = h.nested_sampling(box, pool=n, mc_steps=20, walkers=walkers, seed=100 + seed, min_iterations=250, max_iterations=250)
        assert res["GeneratedNestedSamples"] == 250
        mu = np.abs(res["Points"][np.argsort(res["LogLikelihood"], kind="stable"), 1])     # |mu| of the dead points, in order
        for i in (50, 100, 200):
            devs.append((math.log(mu[i - 1]) + i / n) / (math.sqrt(i) / n))
    devs = np.array(devs).reshape(20, 3)
    assert np.all(np.abs(devs.mean(axis=0)) < 0.6), devs.mean(axis=0)
    assert np.all(devs.std(axis=0, ddof=1) < 2.0)
    h.close()


def test_native_sampler_on_a_gp_matches_grid_integration_and_the_python_driver():
    from bayesianinference_amd import _lib
    X, y = syn.make_dataset(96, 1)
    variables = [("l", 0.05, 1.5), ("sf", 0.2, 3.0), ("sn", 0.03, 0.6)]
    box = np.array([[lo, hi] for _, lo, hi in variables])
    h = _lib.Handle(X, y, "se")
    g = 36
    axes = [lo + (np.arange(g) + 0.5) * (hi - lo) / g for _, lo, hi in variables]
    grid = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, 3)
    vals, info = h.loglik_batch(grid)
    vals = np.where(info == 0, vals, -np.inf)
    want = ns.log_sum_exp(vals) - math.log(len(grid))
    res = h.nested_sampling(box, pool=100, mc_steps=30, walkers=32, seed=11)
    out = ns.evidence_sampling(res, [v[0] for v in variables], 100, np.random.default_rng(0))
    z, se = out["LogEvidence"]["Mean"], out["LogEvidence"]["StandardError"]
    assert abs(z - want) < 4 * se + 0.25, (z, se, want)
    assert res["LikelihoodEvaluations"] >= 100 + 30 * 32
    # a log-uniform prior through prior_kind and the same prior through the callback give the same run (same seed)
    kinds = [1, 1, 1]
    r1 = h.nested_sampling(box, prior_kind=kinds, pool=40, mc_steps=10, walkers=8, seed=5, max_iterations=150, min_iterations=20)
    lognorm = float(np.sum(np.log(np.log(box[:, 1] / box[:, 0]))))
    r2 = h.nested_sampling(box, logprior=lambda th: -float(np.sum(np.log(th))) - lognorm, start=r1["Points"][:40], pool=40,
                           mc_steps=10, walkers=8, seed=5, max_iterations=150, min_iterations=20)
    assert r2["TotalSamples"] > 40 and np.all(np.isfinite(r2["LogPriorPDF"]))
    np.testing.assert_allclose(r2["LogPriorPDF"][:40], r1["LogPriorPDF"][:40], rtol=1e-12)
    with pytest.raises(_lib.GphipError):
        h.nested_sampling(np.array([[1.0, 1.0], [0.2, 3.0], [0.03, 0.6]]))              # lo == hi
    h.close()

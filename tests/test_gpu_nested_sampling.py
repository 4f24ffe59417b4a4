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

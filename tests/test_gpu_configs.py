"""Every BASELINE.json config at its REAL workload on one MI355X (cfg 1-3 live in test_gpu_parity.py:
goldens at N=512, test_f3_scalars at N=8192, test_full_size_properties_n32768).  Here:

  cfg 4  nested sampling over (l, sigma_f, sigma_n), 200 live points x N=4096 log-marginal-likelihoods
         (reference call sites: initial sweep BS:902-916, new-point evaluation BS:1012)
  cfg 5  Matern-5/2 N=65536 d=16 fp32, predictive distribution on 10 000 test points
         (reference: predictFromGaussianProcessInternal BGP:396-422)

cfg 5's log-likelihood scalars are pinned by ONE oracle run at its own size (tests/golden/cfg5_scalars.npz: an in-place LU of
the 34 GB matrix, `oracle/make_golden.py --cfg5`); prediction at N=65536 by the fp64 HIP path on the same data (which IS pinned
against the oracle at every size the oracle finishes) plus size-independent residuals."""
import os

import numpy as np
import pytest

from bayesianinference_amd import _lib, gaussian_process as gp, nested_sampling as ns, synthetic as syn
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


def close(a, b, n, rtol):
    return abs(a - b) <= rtol * max(abs(b), float(n))


def test_cfg4_batch_200_theta_n4096_matches_single_theta_path_and_oracle():
    n, d = 4096, 8
    X, y = syn.make_dataset(n, d)
    Th = syn.theta_batch(200, "se_ard", d)              # log-uniform l, sf in [0.1,10], sn in [0.01,1]
    h = _lib.Handle(X, y, "se_ard")
    out, info = h.loglik_batch(Th)                      # 200 resident workspaces, multi-kernel batched schedule
    assert out.shape == (200,) and info.shape == (200,)
    ok = info == 0
    assert ok.sum() >= 150, info                        # the corners (l=10, sf=10, sn=0.01) are legitimately ill-conditioned
    assert np.all(np.isfinite(out[ok]))
    # all 200 against the single-theta path (one-launch 64-tile dataflow schedule: another summation order)
    single = np.array([h.loglik(th) for th in Th])
    agree = single[:, 1].astype(int) == info
    # the two schedules may give a different verdict only where the smallest pivot sits AT the tolerance
    assert (~agree).sum() <= 4, np.nonzero(~agree)
    both = ok & (single[:, 1] == 0)
    rel = np.abs(out[both] - single[both, 0]) / np.maximum(np.abs(single[both, 0]), n)
    # cond(K) of the batch reaches 1e9+: agreement scales with it; 1e-8 is the bar up to cond 1e8
    cond_bound = 1.0 + n * Th[both, d] ** 2 / Th[both, d + 1] ** 2
    assert np.all(rel <= np.maximum(1e-8, 64 * 2.2e-16 * cond_bound)), (rel.max(), rel.argmax())
    # three well-conditioned thetas against the CPU oracle (LU restatement) at the parity bar
    well = np.nonzero(both & (cond_bound < 1e7))[0][:3]
    assert len(well) == 3
    for i in well:
        want = orc.log_likelihood("se_ard", Th[i], X, y)
        assert close(out[i], want, n, 1e-8), (i, out[i], want)
        assert close(single[i, 0], want, n, 1e-8)
    h.close()


def test_cfg4_nested_sampling_200_live_points_n4096():
    """The workload BASELINE.json cfg 4 names: the sampler itself with SamplePoolSize = 200 on an N=4096
    GP, three hyper-parameters (isotropic SE on d=8 inputs).  Capped at a few dozen iterations; every
    likelihood value the run recorded (initial sweep BS:902-916 and accepted points BS:1012) is
    re-evaluated afterwards by ONE batched call and must reproduce."""
    n, d = 4096, 8
    X, y = syn.make_dataset(n, d)
    variables = [("l", 0.1, 10.0), ("sf", 0.1, 10.0), ("sn", 0.01, 1.0)]
    obj = gp.defineGaussianProcess((X, y), "SE", variables=variables, variablePrior="Uniform")
    assert not obj.failed                                # 100-theta smoke sweep of BS:276-298 passed
    res = ns.nestedSampling(obj, SamplePoolSize=200, MaxIterations=40, MinIterations=40, MonteCarloSteps=6,
                            Walkers=32, Seed=5, PostProcessSamplingRuns=20)
    assert not isinstance(res, str), res
    assert res["SamplePoolSize"] == 200 and res["GeneratedNestedSamples"] == 40 and res["TotalSamples"] == 240
    pts, ll = np.asarray(res["Points"]), np.asarray(res["LogLikelihood"])
    assert pts.shape == (240, 3) and np.all(np.isfinite(ll))
    assert np.all(np.diff(ll) >= 0)                      # calculateWeightsCrude order (BS:818-835)
    handle = obj["GaussianProcessData"]["HIPHandle"]
    again, info = handle.loglik_batch(pts)
    live = ll > gp.MACHINE_LOG_ZERO
    assert np.array_equal(info[live] == 0, np.ones(live.sum(), bool))
    cond_bound = 1.0 + n * pts[:, 1] ** 2 / pts[:, 2] ** 2
    rel = np.abs(again[live] - ll[live]) / np.maximum(np.abs(ll[live]), n)
    assert np.all(rel <= np.maximum(1e-8, 64 * 2.2e-16 * cond_bound[live])), rel.max()
    # the nested samples respect the shrinking likelihood constraint: each new point beat the threshold of its step
    assert np.isfinite(res["LogEvidence"]["Mean"]) and res["LikelihoodEvaluations"] >= 200 + 40
    # one well-conditioned live point against the CPU oracle at the parity bar
    i = int(np.argmin(np.where(live, cond_bound, np.inf)))
    want = orc.log_likelihood("se", pts[i], X, y)
    assert close(ll[i], want, n, 1e-8), (pts[i], ll[i], want)
    handle.close()


def test_cfg5_matern52_n65536_d16_fp32_predict_10k_against_fp64_path():
    n, d, m = 65536, 16, 10000
    X, y = syn.make_dataset(n, d)
    Xs = syn.make_test_points(m, d)
    th = syn.default_theta("matern52_ard", d, dtype="f32")     # l = 1, sf = 1, sn = 0.3
    h32 = _lib.Handle(X, y, "matern52_ard", dtype=32)
    ll32, ld32, qd32, info = h32.loglik_parts(th)
    assert info == 0
    assert h32.fit(th) == 0
    mu32, var32 = h32.predict(Xs)
    assert mu32.shape == (m,) and np.all(np.isfinite(mu32)) and np.all(var32 > 0)
    alpha32 = h32.solve(y)
    h32.close()
    # the same data through the fp64 path (34 GB workspace; pinned against the oracle at smaller N)
    h64 = _lib.Handle(X, y, "matern52_ard", dtype=64)
    ll64, ld64, qd64, info = h64.loglik_parts(th)
    assert info == 0
    assert h64.fit(th) == 0
    mu64, var64 = h64.predict(Xs)
    alpha64 = h64.solve(y)
    h64.close()
    # the ORACLE at cfg 5's own size (one in-place LU of the 34 GB matrix, oracle/make_golden.py --cfg5): fp64 HIP at 1e-8,
    # fp32 HIP at the stated 1e-3
    gpath = os.path.join(os.path.dirname(__file__), "golden", "cfg5_scalars.npz")
    if os.path.exists(gpath):
        gold = np.load(gpath)
        assert int(gold["n"]) == n and int(gold["d"]) == d and int(gold["info"]) == 0
        assert abs(float(gold["xsum"]) - float(X.sum())) < 1e-6 and abs(float(gold["ysum"]) - float(y.sum())) < 1e-6
        np.testing.assert_allclose(gold["theta"], th)
        assert close(ll64, float(gold["loglik"]), n, 1e-8) and close(ld64, float(gold["logdet"]), n, 1e-8)
        assert close(qd64, float(gold["quad"]), n, 1e-8)
        assert close(ll32, float(gold["loglik"]), n, 1e-3) and close(ld32, float(gold["logdet"]), n, 1e-3)
        assert close(qd32, float(gold["quad"]), n, 1e-3)
        # ... and the PREDICTION (BGP:396-422) at cfg 5's own size: the oracle's mu*, sigma* from the same LU at the first 64 of
        # the 10 000 test points -- fp64 HIP at 1e-7, fp32 HIP at the stated 1e-3 (round 3 compared HIP with HIP here)
        if "mu" in gold.files:
            k = len(gold["mu"])
            np.testing.assert_allclose(gold["Xs"], Xs[:k])
            np.testing.assert_allclose(mu64[:k], gold["mu"], rtol=1e-7, atol=1e-9)
            np.testing.assert_allclose(np.sqrt(var64[:k]), gold["sd"], rtol=1e-7)
            np.testing.assert_allclose(mu32[:k], gold["mu"], rtol=1e-3, atol=1e-3)
            np.testing.assert_allclose(np.sqrt(var32[:k]), gold["sd"], rtol=1e-3)
    # stated fp32 tolerance: 1e-3 relative
    assert abs(ld32 - ld64) <= 1e-3 * max(abs(ld64), n)
    assert abs(qd32 - qd64) <= 1e-3 * max(abs(qd64), n)
    assert abs(ll32 - ll64) <= 1e-3 * max(abs(ll64), n)
    np.testing.assert_allclose(mu32, mu64, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(np.sqrt(var32), np.sqrt(var64), rtol=1e-3)
    # size-independent properties: the quadratic form equals y . K^-1 y from an independent forward +
    # backward solve, and sampled rows of K (K^-1 y) give y back
    assert close(float(y @ alpha64), qd64, n, 1e-9)
    assert close(float(y @ alpha32), qd32, n, 1e-3)
    idx = np.array([0, 1, 127, 128, 32768, 40001, 65535])
    ell, sf, sn, _ = orc.split_theta("matern52_ard", d, th)
    Krows = orc.kernel_matrix("matern52_ard", ell, sf, X[idx], X)
    Krows[np.arange(len(idx)), idx] += sn * sn
    np.testing.assert_allclose(Krows @ alpha64, y[idx], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(Krows @ alpha32, y[idx], rtol=2e-3, atol=2e-3)
    # far from the data the prediction tends to the prior: mu* -> m(x*) = 0, var* -> sf^2 + sn^2 (BGP:407-417)
    far = np.full((3, d), 50.0)
    h = _lib.Handle(X[:2048], y[:2048], "matern52_ard", dtype=32)
    assert h.fit(th) == 0
    mu_far, var_far = h.predict(far)
    np.testing.assert_allclose(mu_far, 0.0, atol=1e-6)
    np.testing.assert_allclose(var_far, 1.0 + 0.09, rtol=1e-5)
    h.close()

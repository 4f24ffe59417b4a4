"""Pins the CPU oracle (oracle/gp_oracle.py): closed forms from BGP:181-199, the MVN second
formulation (BGP:273-292), 50-digit mpmath, and the committed golden vectors."""
import math
import os

import numpy as np
import pytest

from bayesianinference_amd import synthetic as syn
from oracle import gp_oracle as orc

L2PI = math.log(2 * math.pi)


def test_closed_form_n1():
    x, y, ell, sf, sn, mu = 0.3, 1.7, 0.8, 1.3, 0.4, 0.25
    v = sf * sf + sn * sn
    want = -0.5 * (L2PI + math.log(v) + (y - mu) ** 2 / v)
    got = orc.log_likelihood("se", [ell, sf, sn, mu], [[x]], [y], mean="const")
    assert got == pytest.approx(want, rel=1e-14)


def test_closed_form_n2():
    X = np.array([[0.1, -0.4], [0.7, 0.2]])
    y = np.array([0.3, -1.1])
    ell, sf, sn = np.array([0.9, 1.4]), 1.2, 0.3
    r2 = np.sum(((X[0] - X[1]) / ell) ** 2)
    a = sf * sf + sn * sn
    b = sf * sf * math.exp(-0.5 * r2)
    det = a * a - b * b
    quad = (a * y[0] ** 2 - 2 * b * y[0] * y[1] + a * y[1] ** 2) / det
    want = -0.5 * (2 * L2PI + math.log(det) + quad)
    got = orc.log_likelihood("se_ard", [*ell, sf, sn], X, y)
    assert got == pytest.approx(want, rel=1e-13)


def test_null_kernel_is_independent_normals():
    X, y = syn.make_dataset(40, 2)
    sn, mu = 0.7, 0.1
    want = float(np.sum(-0.5 * (L2PI + math.log(sn * sn) + (y - mu) ** 2 / (sn * sn))))
    got = orc.log_likelihood("null", [sn, mu], X, y, mean="const")
    assert got == pytest.approx(want, rel=1e-13)


@pytest.mark.parametrize("kernel,d", [("se", 1), ("se_ard", 4), ("matern52_ard", 3), ("matern52", 2)])
def test_lu_restatement_matches_mvn_formulation(kernel, d):
    X, y = syn.make_dataset(200, d)
    for th in syn.theta_batch(4, kernel, d):
        th[-1] = max(th[-1], 0.05)
        a = orc.log_likelihood(kernel, th, X, y)
        b = orc.log_likelihood_mvn(kernel, th, X, y)
        assert a == pytest.approx(b, rel=1e-10)


def test_prediction_limits():
    X, y = syn.make_dataset(50, 1)
    th = np.array([0.2, 1.1, 1e-4])
    far = np.array([[50.0]])
    mu, sd = orc.predict_internal("se", th, X, y, far)
    assert abs(mu[0]) < 1e-12
    assert sd[0] ** 2 == pytest.approx(th[1] ** 2 + th[2] ** 2, rel=1e-12)
    # short length-scale: inputs decouple, so mu*(x_i) -> y_i sf^2/(sf^2+sn^2) -> y_i as nugget -> 0
    X, y = syn.make_dataset(20, 1)
    mu, sd = orc.predict_internal("se", np.array([1e-4, 1.1, 1e-4]), X, y, X[:5])
    np.testing.assert_allclose(mu, y[:5], atol=1e-6)


def test_against_mpmath_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "hp_mpmath.npz"))
    for key, kernel in (("se_ard_n24", "se_ard"), ("matern52_ard_n48", "matern52_ard"), ("se_n32", "se")):
        X, y, th = g[f"{key}_X"], g[f"{key}_y"], g[f"{key}_theta"]
        ll, ld, qd, info = orc.log_likelihood(kernel, th, X, y, parts=True)
        assert info == 0
        assert ll == pytest.approx(float(g[f"{key}_loglik"]), rel=1e-12)
        assert ld == pytest.approx(float(g[f"{key}_logdet"]), rel=1e-12)
        assert qd == pytest.approx(float(g[f"{key}_quad"]), rel=1e-11)
        mu, sd = orc.predict_internal(kernel, th, X, y, g[f"{key}_Xs"])
        np.testing.assert_allclose(mu, g[f"{key}_mu"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(sd, g[f"{key}_sd"], rtol=1e-10)


def test_oracle_against_mpmath_at_cfg1_size(golden_dir):
    """The LU restatement against 30-digit arithmetic at cfg 1's own size (N=512, d=1; oracle/make_golden.py fhp)."""
    g = np.load(os.path.join(golden_dir, "hp_mpmath.npz"))
    key = "se_n512"
    X, y, th = g[f"{key}_X"], g[f"{key}_y"], g[f"{key}_theta"]
    ll, ld, qd, info = orc.log_likelihood("se", th, X, y, parts=True)
    assert info == 0
    assert ll == pytest.approx(float(g[f"{key}_loglik"]), rel=1e-11)
    assert ld == pytest.approx(float(g[f"{key}_logdet"]), rel=1e-11)
    assert qd == pytest.approx(float(g[f"{key}_quad"]), rel=1e-10)
    mu, sd = orc.predict_internal("se", th, X, y, g[f"{key}_Xs"])
    np.testing.assert_allclose(mu, g[f"{key}_mu"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(sd, g[f"{key}_sd"], rtol=1e-9)
    # and the committed F1 fixture's first theta is this very problem
    f1 = np.load(os.path.join(golden_dir, "f1_se_n512_d1.npz"))
    assert np.array_equal(f1["thetas"][0], th) and float(f1["loglik"][0]) == pytest.approx(float(g[f"{key}_loglik"]), rel=1e-11)


def test_mpmath_live_small():
    from oracle import hp_oracle as hp
    X, y = syn.make_dataset(12, 2)
    th = [0.6, 1.4, 0.9, 0.15]
    ll, ld, qd = hp.log_likelihood("se_ard", th, X.tolist(), y.tolist())
    assert orc.log_likelihood("se_ard", th, X, y) == pytest.approx(ll, rel=1e-13)


@pytest.mark.parametrize("kernel,family,theta", [("matern32_ard", "matern32", [0.7, 1.3, 0.9, 0.2]), ("rq_ard", "rq", [0.7, 1.3, 1.8, 0.9, 0.2]),
                                                 ("matern32", "matern32", [0.8, 1.1, 0.25]), ("rq", "rq", [0.8, 0.6, 1.1, 0.25]),
                                                 ("se + const", "se", [0.8, 1.1, 0.5, 0.25]), ("rq_ard + const", "rq", [0.7, 1.3, 2.5, 0.9, 0.4, 0.2])])
def test_general_kernel_forms_against_mpmath(kernel, family, theta):
    """Round 6: the families that got a matrix-pipe build on the device (Matern-3/2, rational quadratic, `term + const` -- the
    reference's own example kernel is c + SE, BGP:16) pinned by an independent 50-digit evaluation of the same formulas:
    gp_oracle's general grammar (parse_kernel / split_general / general_kernel_matrix) against oracle/hp_oracle.py."""
    from oracle import hp_oracle as hp
    d = 2
    X, y = syn.make_dataset(14, d)
    terms, op, c, sn, mu = orc.split_general(kernel, d, theta)
    (_, ell, alpha, sf), = terms
    ll, ld, qd = hp.general_log_likelihood(family, ell, sf, sn, X.tolist(), y.tolist(), alpha=alpha, offset=c)
    got, ld_o, qd_o, info = orc.log_likelihood(kernel, theta, X, y, parts=True)
    assert info == 0
    assert got == pytest.approx(ll, rel=1e-12) and ld_o == pytest.approx(ld, rel=1e-12, abs=1e-12) and qd_o == pytest.approx(qd, rel=1e-11)


@pytest.mark.parametrize("name", ["f1_se_n512_d1", "f2_se_ard_n256_d8", "f2_matern52_ard_n256_d8",
                                  "f2_matern52_const_n333_d3"])
def test_oracle_reproduces_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    kernel, mean = str(g["kernel"]), str(g["mean"])
    for i in (0, 5):
        ll = orc.log_likelihood(kernel, g["thetas"][i], g["X"], g["y"], mean)
        assert ll == pytest.approx(float(g["loglik"][i]), rel=1e-12)
    mu, sd = orc.predict_internal(kernel, g["thetas"][1], g["X"], g["y"], g["Xs"], mean)
    np.testing.assert_allclose(mu, g["pred_mu"][1], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(sd, g["pred_sd"][1], rtol=1e-9)


def test_sentinel_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "f4_sentinel.npz"))
    assert int(g["dup_info"]) == 1 and float(g["dup_loglik"]) == orc.MACHINE_LOG_ZERO
    assert int(g["ill_info"]) == 1
    assert int(g["ok_info"]) == 0 and np.isfinite(float(g["ok_loglik"]))
    with pytest.warns(Warning):
        assert orc.log_likelihood("se_ard", g["dup_theta"], g["dup_X"], g["dup_y"]) == orc.MACHINE_LOG_ZERO


def test_synthetic_generator_pinned(golden_dir):
    g = np.load(os.path.join(golden_dir, "f3_scalars.npz"))
    X, y = syn.make_dataset(2048, 8)
    assert float(X.sum()) == float(g["xsum"][0]) and float(y.sum()) == float(g["ysum"][0])
    # slices regenerate independently (multi-GPU ranks rebuild their own rows)
    np.testing.assert_array_equal(syn.make_inputs(10, 8, row0=100), X[100:110])
    np.testing.assert_array_equal(syn.make_outputs(X[100:110], row0=100), y[100:110])


def test_log_space_helpers():
    assert orc.log_sum_exp([-np.inf, 0.0, math.log(3.0)]) == pytest.approx(math.log(4.0))
    assert orc.log_sum_exp([-np.inf]) == -np.inf
    assert orc.log_add(math.log(2.0), math.log(3.0)) == pytest.approx(math.log(5.0))
    assert orc.log_subtract(math.log(5.0), math.log(3.0)) == pytest.approx(math.log(2.0))


@pytest.mark.parametrize("kernel,d,mean", [("se", 1, "zero"), ("se_ard", 3, "const"), ("matern52_ard", 2, "zero"),
                                           ("matern52", 2, "const")])
def test_gradient_oracle_matches_finite_differences(kernel, d, mean):
    X, y = syn.make_dataset(60, d)
    th = syn.default_theta(kernel, d)
    th[-1] = 0.25
    if mean == "const":
        th = np.append(th, 0.3)
    g = orc.log_likelihood_grad(kernel, th, X, y, mean)
    for i in range(len(th)):
        e = np.zeros_like(th)
        e[i] = 1e-6 * max(1.0, abs(th[i]))
        fd = (orc.log_likelihood(kernel, th + e, X, y, mean) - orc.log_likelihood(kernel, th - e, X, y, mean)) / (2 * e[i])
        assert g[i] == pytest.approx(fd, rel=2e-5, abs=1e-6)


def _reference_fixtures(golden_dir):
    import glob
    import json
    files = sorted(f for f in glob.glob(os.path.join(golden_dir, "reference_*.json"))
                   if not f.endswith("reference_inputs.json"))
    inputs = {c["name"]: c for c in json.load(open(os.path.join(golden_dir, "reference_inputs.json")))["cases"]}
    return [(json.load(open(f)), inputs) for f in files]


def test_reference_inputs_match_the_committed_goldens(golden_dir):
    """The hand-over file of oracle/make_reference_golden.wl carries exactly the golden fixtures' inputs."""
    import json
    cases = {c["name"]: c for c in json.load(open(os.path.join(golden_dir, "reference_inputs.json")))["cases"]}
    g = np.load(os.path.join(golden_dir, "f2_matern52_ard_n256_d8.npz"))
    c = cases["f2_matern52_ard_n256_d8"]
    assert np.array_equal(np.array(c["X"]), g["X"]) and np.array_equal(np.array(c["y"]), g["y"])
    assert np.array_equal(np.array(c["thetas"]), g["thetas"]) and c["kernel"] == "matern52_ard"
    assert set(cases) >= {"f1_se_n96_d1", "f1_se_n512_d1", "f2_se_ard_n256_d8", "f2_matern52_const_n333_d3",
                          "f4_dup", "f4_ill", "f4_ok"}


def test_oracle_matches_reference_fixture(golden_dir):
    """The pin: outputs of the REAL reference (oracle/make_reference_golden.wl, needs a Wolfram kernel) against the
    oracle at 1e-10 relative.  No such file can be produced in the build containers; until one is committed the
    oracle is pinned only by closed forms / MVN / mpmath and parity is reported as UNPINNED."""
    fixtures = _reference_fixtures(golden_dir)
    if not fixtures:
        pytest.skip("parity unpinned: no reference-produced fixture (tests/golden/reference_<case>.json) present -- "
                    "run `wolframscript -file oracle/make_reference_golden.wl <BayesianInference dir>`")
    for ref, inputs in fixtures:
        c = inputs[ref["name"]]
        X, y, Xs = np.array(c["X"]), np.array(c["y"]), np.array(c["Xs"])
        n = len(y)
        for i, th in enumerate(np.array(ref["thetas"])):
            ll, ld, qd, info = orc.log_likelihood(c["kernel"], th, X, y, c["mean"], parts=True)
            if ref["loglik_is_sentinel"][i]:
                assert info != 0, (ref["name"], i)
                continue
            assert info == 0
            assert abs(ll - ref["loglik"][i]) <= 1e-10 * max(abs(ref["loglik"][i]), n), (ref["name"], i)
            if ref["logdet"][i] is not None:
                assert abs(ld - ref["logdet"][i]) <= 1e-10 * max(abs(ref["logdet"][i]), n)
                assert abs(qd - ref["quad"][i]) <= 1e-9 * max(abs(ref["quad"][i]), n)
        for i, (mu, sd) in enumerate(zip(ref["pred_mu"], ref["pred_sd"])):
            pts = np.array(ref["pred_points"])             # DeleteDuplicates'd inputs, in the reference's order
            mo, so = orc.predict_internal(c["kernel"], np.array(ref["thetas"])[i], X, y, pts, c["mean"])
            np.testing.assert_allclose(mo, mu, rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(so, sd, rtol=1e-9)


def test_point_dependent_nugget_and_mean_functions_reduce_to_the_constant_forms():
    """nugget_fn / mean_fn (BGP:37, 300) with constant functions are the default path; a heteroscedastic nugget changes K's
    diagonal only; kappa carries the test-point nugget (BGP:113)."""
    X, y = syn.make_dataset(40, 2)
    Xs = syn.make_test_points(5, 2)
    th = np.array([0.7, 1.1, 1.2, 0.3, 0.25])
    base = orc.log_likelihood("se_ard", th, X, y, "const")
    same = orc.log_likelihood("se_ard", th, X, y, "const", nugget_fn=lambda x: 0.09, mean_fn=lambda x: 0.25)
    assert same == pytest.approx(base, rel=1e-14)
    nf = lambda x: 0.09 * (1 + x[0] ** 2)                     # noqa: E731
    K0 = orc.covariance_matrix("se_ard", th, X, "const")
    K1 = orc.covariance_matrix("se_ard", th, X, "const", nugget_fn=nf)
    off = ~np.eye(40, dtype=bool)
    assert np.array_equal(K0[off], K1[off])
    np.testing.assert_allclose(np.diag(K1) - np.diag(K0), 0.09 * X[:, 0] ** 2, rtol=1e-12, atol=1e-15)
    _, kappa = orc.k_and_kappa("se_ard", th, X, Xs, "const", nugget_fn=nf)
    np.testing.assert_allclose(kappa, 1.2 ** 2 + 0.09 * (1 + Xs[:, 0] ** 2))
    mu, sd = orc.predict_internal("se_ard", th, X, y, Xs, "const", nugget_fn=lambda x: 0.09, mean_fn=lambda x: 0.25)
    mu0, sd0 = orc.predict_internal("se_ard", th, X, y, Xs, "const")
    np.testing.assert_allclose(mu, mu0, rtol=1e-13)
    np.testing.assert_allclose(sd, sd0, rtol=1e-13)

"""Nested-sampling driver feeding the GP likelihood path in batches.

Host-side restatement of the reference's sampler (BayesianStatistics.wl, cited BS:line):

  calculateXValues["Log"]        BS:790-802     log X_i = -i/n, live points (i/(n+1)) e^{-nDel/n}
  trapezoidWeigths["Log"]        BS:757-771     trapezoid weights with reflected end points
  calculateWeightsCrude          BS:818-835     sort by (LogLikelihood, Point), weights + LogLikelihood
  calculateEntropy               BS:804-816
  nestedSamplingInternal         BS:859-1040    main loop, stop rule BS:967-978
  nsMCMC / nsDensity             BS:707-745, 602-628   constrained-prior Metropolis
  evidenceSampling               BS:1158-1291   Monte-Carlo resampling of the X values
  combineRuns                    BS:1293-1315
  parallelNestedSampling         BS:1320-1371   independent replicas, merged by likelihood order

MI355X-first difference (documented): the reference advances ONE Adaptive-Metropolis
chain for "MonteCarloSteps" strictly sequential likelihood calls per iteration (BS:990-1004).
Here `Walkers` chains start from random live points and advance in lock-step, so every MCMC step is
ONE batched likelihood call theta[W x p] -> l[W] (gphip_loglik_batch).  Each walker's end point is
a draw from the prior restricted to L > L*_i; it is consumed at a later iteration i+j only if it
still satisfies L > L*_{i+j} (exact rejection step), so the X-shrinkage law log X_i = -i/n of the
reference is preserved.  The reference's chain kernel (Statistics`MCMC`, closed source) cannot be
reproduced bit-for-bit; only log Z +- its own standard error is comparable.
"""
from __future__ import annotations

import math

import numpy as np

MACHINE_LOG_ZERO = -1.7976931348623157e308

DEFAULTS = {                                   # BS:833-855
    "SamplePoolSize": 100, "MaxIterations": 10000, "MinIterations": 100, "MonteCarloSteps": 200,
    "TerminationFraction": 0.01, "PostProcessSamplingRuns": 100, "Walkers": 32, "Seed": 0,
    "MinMaxAcceptanceRate": (0.0, 1.0),
}


# ---------------------------------------------------------------------------------------------
# deterministic pieces (pinned by hand in tests/test_nested_sampling.py)
# ---------------------------------------------------------------------------------------------
def log_subtract(logy, logx):                   # BU:337-343
    return logy + np.log1p(-np.exp(np.subtract(logx, logy)))


def log_add(logy, logx):                        # BU:345-356
    hi, lo = np.maximum(logx, logy), np.minimum(logx, logy)
    return hi + np.log1p(np.exp(lo - hi))


def log_sum_exp(v) -> float:                    # BU:318-335 (drops -Infinity)
    v = np.asarray(v, dtype=np.float64)
    v = v[v > -np.inf]
    if v.size == 0:
        return -np.inf
    m = v.max()
    return float(m + np.log(np.sum(np.exp(v - m))))


def calculate_x_values_log(n_pool: int, n_deleted: int) -> np.ndarray:
    """BS:790-802: deleted points -k/n (k=1..nDel); live points log(i/(n+1)) - nDel/n, i = n..1."""
    dead = -np.arange(1, n_deleted + 1) / n_pool
    live = np.log(np.arange(n_pool, 0, -1) / (n_pool + 1.0)) - n_deleted / n_pool
    return np.concatenate([dead, live])


def trapezoid_weights_log(logx: np.ndarray) -> np.ndarray:
    """BS:757-771: w_i = (X_{i-1} - X_{i+1})/2 with X_0 = 2 - X_1 and the last weight (X_{m-1}+X_m)/2."""
    logx = np.asarray(logx, dtype=np.float64)
    left = np.concatenate([[log_subtract(math.log(2.0), logx[0])], logx[:-2]])
    inner = log_subtract(left, logx[1:])
    last = log_add(logx[-2], logx[-1])
    return math.log(0.5) + np.concatenate([inner, [last]])


def calculate_weights_crude(points: np.ndarray, loglik: np.ndarray, n_pool: int):
    """BS:818-835.  Returns (order, logX, crude log posterior weights) with `order` sorting the
    samples by (LogLikelihood, Point) -- ties broken by the point, as the reference does."""
    order = np.lexsort(tuple(points[:, j] for j in range(points.shape[1] - 1, -1, -1)) + (loglik,))
    logx = calculate_x_values_log(n_pool, len(loglik) - n_pool)
    return order, logx, trapezoid_weights_log(logx) + loglik[order]


def calculate_entropy(log_weights, loglik, log_evidence) -> float:
    """BS:804-816: sum_i (w_i/Z) log L_i - log Z."""
    ll = np.where(np.isfinite(loglik), loglik, 0.0)
    return float(np.exp(log_weights - log_evidence) @ ll - log_evidence)


# ---------------------------------------------------------------------------------------------
# batched likelihood adapter
# ---------------------------------------------------------------------------------------------
def _batched(fn):
    def call(thetas):
        thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
        try:
            out = np.asarray(fn(thetas), dtype=np.float64)
            if out.shape == (len(thetas),):
                return out
        except (TypeError, ValueError):
            pass
        return np.array([fn(t) for t in thetas], dtype=np.float64)
    return call


def _in_box(points, lo, hi):
    return np.all((points >= lo) & (points <= hi), axis=1)


def constrained_walkers(loglik_batch, logprior, live_points, threshold, cov, lo, hi, n_walkers, n_steps, rng):
    """The batched counterpart of nsMCMC + nsDensity (BS:707-745, 619-628): random-walk Metropolis on
    the prior restricted to {L > threshold, theta in box}, W walkers in lock-step, proposal covariance
    (2.38^2/d) cov (Haario et al. adaptive-Metropolis scaling, with cov re-estimated from the live
    points every iteration like covEst at BS:988).  Returns (points, logliks, acceptance rate)."""
    d = live_points.shape[1]
    start = rng.integers(0, len(live_points), n_walkers)          # RandomChoice[bestPoints], BS:992
    x = live_points[start].copy()
    lx = np.full(n_walkers, np.nan)
    lp = np.array([logprior(t) for t in x])
    chol = np.linalg.cholesky((2.38 ** 2 / d) * cov + 1e-12 * np.diag(np.diag(cov) + 1e-300))
    accepted = 0
    for _ in range(n_steps):
        prop = x + rng.standard_normal((n_walkers, d)) @ chol.T
        ok = _in_box(prop, lo, hi)
        lpp = np.array([logprior(t) if o else MACHINE_LOG_ZERO for t, o in zip(prop, ok)])
        ok &= lpp > MACHINE_LOG_ZERO
        ll = np.full(n_walkers, MACHINE_LOG_ZERO)
        if ok.any():
            ll[ok] = loglik_batch(prop[ok])                       # ONE batched likelihood call per step
        acc = ok & (ll > threshold) & (np.log(rng.random(n_walkers)) < lpp - lp)
        x[acc], lx[acc], lp[acc] = prop[acc], ll[acc], lpp[acc]
        accepted += int(acc.sum())
    moved = ~np.isnan(lx)                                          # walkers that never moved duplicate a live point
    return x[moved], lx[moved], accepted / float(n_walkers * n_steps)


# ---------------------------------------------------------------------------------------------
# BS:859-1040 nestedSamplingInternal
# ---------------------------------------------------------------------------------------------
def nested_sampling_internal(loglik, logprior, starting_points, params, **opts):
    o = {**DEFAULTS, **opts}
    rng = np.random.default_rng(o["Seed"])
    loglik_batch = _batched(loglik)
    lo = np.array([p[1] for p in params], dtype=np.float64)
    hi = np.array([p[2] for p in params], dtype=np.float64)
    pts = np.atleast_2d(np.asarray(starting_points, dtype=np.float64))
    n = len(pts)
    ll = loglik_batch(pts)                                        # initial sweep, BS:902-916: ONE batched call
    if not np.all(np.isfinite(ll)):
        return "Bad likelihood function"                          # BS:917-921
    lpr = np.array([logprior(t) for t in pts])
    acc_rates = [np.nan] * n
    cov = np.atleast_2d(np.cov(pts.T))                            # BS:923
    max_it = max(o["MaxIterations"], o["MinIterations"])
    min_it = min(o["MaxIterations"], o["MinIterations"])
    log_evidence, entropy, iteration = MACHINE_LOG_ZERO, 0.0, 1
    cand_pts = np.zeros((0, pts.shape[1]))
    cand_ll = np.zeros(0)
    cand_rate = np.zeros(0)                                       # acceptance rate of the walker batch a candidate came from
    n_evals = n
    while iteration <= max_it:
        if iteration > 1 and iteration > min_it:                  # stop rule BS:967-978
            order, logx, logw = calculate_weights_crude(pts, ll, n)
            missing = math.exp(logx.min()) * math.exp(min(ll.max() - log_evidence, 700.0))
            if missing <= o["TerminationFraction"]:
                break
        best = np.argsort(ll, kind="stable")[-n:]                 # bestPoints, BS:980
        threshold = ll[best].min()
        cov = 0.5 * (cov + np.atleast_2d(np.cov(pts[best].T)))    # BS:988
        keep = cand_ll > threshold                                # exact rejection of stale candidates
        cand_pts, cand_ll, cand_rate = cand_pts[keep], cand_ll[keep], cand_rate[keep]
        factor = 1.0
        rmin, rmax = o["MinMaxAcceptanceRate"]
        while len(cand_ll) == 0:
            steps = int(math.ceil(factor * o["MonteCarloSteps"]))
            cand_pts, cand_ll, rate = constrained_walkers(loglik_batch, logprior, pts[best], threshold, cov,
                                                          lo, hi, o["Walkers"], steps, rng)
            n_evals += o["Walkers"] * steps
            # BS:990-1004: the chain is re-run with 1.25x the steps until its acceptance rate lies inside
            # "MinMaxAcceptanceRate" (Between[rate, {min, max}], default {0, 1} = always); here the rate is that of
            # the whole walker batch, and a batch outside the window is discarded like the reference's chain
            cand_rate = np.full(len(cand_ll), rate)
            if not (rmin <= rate <= rmax):
                cand_pts, cand_ll, cand_rate = cand_pts[:0], cand_ll[:0], cand_rate[:0]
            factor *= 1.25                                        # BS:1003 step inflation on failure
            if factor > 50:
                return "Bad likelihood function"
        pts = np.vstack([pts, cand_pts[:1]])
        ll = np.append(ll, cand_ll[0])                            # BS:1012 (value carried from the chain)
        lpr = np.append(lpr, logprior(cand_pts[0]))
        acc_rates.append(cand_rate[0])
        cand_pts, cand_ll, cand_rate = cand_pts[1:], cand_ll[1:], cand_rate[1:]
        order, logx, logw = calculate_weights_crude(pts, ll, n)
        log_evidence = log_sum_exp(logw)                          # BS:1019
        entropy = calculate_entropy(logw, ll[order], log_evidence)
        iteration += 1
    result = {
        "Points": pts, "LogLikelihood": ll, "LogPriorPDF": lpr, "AcceptanceRate": np.array(acc_rates),
        "SamplePoolSize": n, "GeneratedNestedSamples": len(ll) - n, "TotalSamples": len(ll),
        "ParameterRanges": np.stack([pts.min(axis=0), pts.max(axis=0)], axis=1),
        "LikelihoodEvaluations": n_evals, "Seed": o["Seed"],
    }
    return evidence_sampling(result, [p[0] for p in params], o["PostProcessSamplingRuns"], rng)


# ---------------------------------------------------------------------------------------------
# BS:1158-1291 evidenceSampling
# ---------------------------------------------------------------------------------------------
def evidence_sampling(result: dict, param_names, n_runs=100, rng=None) -> dict:
    rng = rng or np.random.default_rng(0)
    pts, ll, n = result["Points"], result["LogLikelihood"], result["SamplePoolSize"]
    order, logx, logw = calculate_weights_crude(pts, ll, n)
    pts, ll = pts[order], ll[order]
    crude_logz = log_sum_exp(logw)
    out = dict(result)
    for key in ("LogPriorPDF", "AcceptanceRate"):
        if key in out and len(out[key]) == len(order):
            out[key] = np.asarray(out[key])[order]
    out.update({
        "Points": pts, "LogLikelihood": ll, "LogX": logx, "X": np.exp(logx),
        "CrudeLogEvidence": crude_logz, "LogLikelihoodMaximum": float(ll.max()),
        "LogEstimatedMissingEvidence": float(logx.min() + ll.max()),
        "CrudeRelativeEntropy": calculate_entropy(logw, ll, crude_logz),
        "CrudeLogPosteriorWeight": logw - crude_logz, "CrudePosteriorWeight": np.exp(logw - crude_logz),
    })
    n_del = len(ll) - n
    if n_runs and n_runs > 0:
        # BS:1200-1221: log X of deleted points = -cumsum Exp(n) draws; live points: sorted draws of
        # -(shifted Exp(1)) below the last deleted X
        steps = -rng.exponential(1.0 / n, size=(n_runs, n_del))
        dead = np.cumsum(steps, axis=1) if n_del else np.zeros((n_runs, 0))
        floor = dead[:, -1] if n_del else np.zeros(n_runs)
        live = -np.sort(-(floor[:, None] - rng.exponential(1.0, size=(n_runs, n))), axis=1)
        sampled_logx = np.concatenate([dead, live], axis=1)
        logw_runs = np.array([trapezoid_weights_log(row) for row in sampled_logx]) + ll[None, :]
        z = np.array([log_sum_exp(r) for r in logw_runs])
        post = np.exp(logw_runs - z[:, None])
        psamples = post @ pts
        ll0 = np.where(np.isfinite(ll), ll, 0.0)
        out.update({
            "LogEvidence": {"Mean": float(z.mean()), "StandardError": float(z.std(ddof=1))},
            "ParameterExpectedValues": {nm: {"Mean": float(psamples[:, j].mean()),
                                             "StandardError": float(psamples[:, j].std(ddof=1))}
                                        for j, nm in enumerate(param_names)},
            "RelativeEntropy": {"Mean": float((post @ ll0 - z).mean()),
                                "StandardError": float((post @ ll0 - z).std(ddof=1))},
        })
    # "Samples" in the reference's shape: sorted by decreasing posterior weight (BS:1240)
    by_w = np.argsort(-out["CrudeLogPosteriorWeight"], kind="stable")
    out["Samples"] = [{"Point": pts[i], "LogLikelihood": float(ll[i]), "X": float(out["X"][i]),
                       "LogX": float(logx[i]), "CrudeLogPosteriorWeight": float(out["CrudeLogPosteriorWeight"][i]),
                       "CrudePosteriorWeight": float(out["CrudePosteriorWeight"][i])} for i in by_w]
    if "LogPriorPDF" in out and len(out["LogPriorPDF"]) == len(ll):          # BS:884-886 keeps it per sample
        for smp, i in zip(out["Samples"], by_w):
            smp["LogPriorPDF"] = float(out["LogPriorPDF"][i])
    return out


def combine_runs(results, param_names, n_runs=100, rng=None) -> dict:
    """BS:1293-1315: merge the samples of independent runs (duplicates by point removed), pool sizes add."""
    pts = np.vstack([r["Points"] for r in results])
    ll = np.concatenate([r["LogLikelihood"] for r in results])
    _, first = np.unique(pts, axis=0, return_index=True)
    first = np.sort(first)
    merged = {"Points": pts[first], "LogLikelihood": ll[first],
              "SamplePoolSize": int(sum(r["SamplePoolSize"] for r in results))}
    merged["GeneratedNestedSamples"] = len(first) - merged["SamplePoolSize"]
    merged["TotalSamples"] = len(first)
    return evidence_sampling(merged, param_names, n_runs, rng)


# ---------------------------------------------------------------------------------------------
# public entry points on inferenceObject (BS:1099-1136, 1320-1371)
# ---------------------------------------------------------------------------------------------
def generate_starting_points(obj, n, rng):
    """BS:1046-1068 with a box prior: RandomVariate[prior, n] for "Uniform" / product priors; otherwise
    rejection from the box against the log prior."""
    params = obj["Parameters"]
    lo = np.array([p[1] for p in params], dtype=np.float64)
    hi = np.array([p[2] for p in params], dtype=np.float64)
    prior = obj["PriorDistribution"] if "PriorDistribution" in obj else "Uniform"
    if isinstance(prior, (list, tuple)):
        cols = []
        for dist, a, b in zip(prior, lo, hi):
            ua, ub = dist.cdf(a), dist.cdf(b)
            cols.append(dist.ppf(ua + rng.random(n) * (ub - ua)))
        return np.column_stack(cols)
    return lo + rng.random((n, len(params))) * (hi - lo)


def nestedSampling(obj, **opts):
    """BS:1099-1136: returns the object joined with the sampling result (obj.append(result))."""
    from .gaussian_process import inferenceObject
    o = {**DEFAULTS, **opts}
    rng = np.random.default_rng(o["Seed"] + 7919)
    start = opts.get("StartingPoints")
    if start is None:
        start = generate_starting_points(obj, o["SamplePoolSize"], rng)
    res = nested_sampling_internal(obj["LogLikelihoodFunction"], obj["LogPriorPDFFunction"], start,
                                   obj["Parameters"], **{k: v for k, v in o.items() if k != "StartingPoints"})
    if isinstance(res, str):
        return res
    return inferenceObject(obj).append({**res, "StartingPoints": np.asarray(start)})


def parallelNestedSampling(obj, ParallelRuns=4, dist=None, **opts):
    """BS:1320-1371: `ParallelRuns` independent replicas, merged with combineRuns.  With
    torch.distributed initialised the replicas are dealt to ranks (one GPU each, no data-path
    collective) and the merged result is formed on every rank from an all_gather of the sample lists."""
    from .gaussian_process import inferenceObject
    seeds = [opts.get("Seed", 0) + 1000 * (r + 1) for r in range(ParallelRuns)]
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None and dist.is_initialized() else (0, 1)
    mine = []
    for r in range(rank, ParallelRuns, world):
        res = nestedSampling(obj, **{**opts, "Seed": seeds[r], "PostProcessSamplingRuns": 0})
        if isinstance(res, str):
            return res
        mine.append({k: res[k] for k in ("Points", "LogLikelihood", "SamplePoolSize")})
    runs = mine
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        runs = [r for part in gathered for r in part]
    merged = combine_runs(runs, [p[0] for p in obj["Parameters"]], opts.get("PostProcessSamplingRuns", 100),
                          np.random.default_rng(opts.get("Seed", 0)))
    return inferenceObject(obj).append(merged)


def inferenceObject_take(obj, n_samples: int):
    """The object restricted to its `n_samples` heaviest posterior samples (weights renormalised by the
    consumer): predictFromGaussianProcess factors K once per sample (BGP:355-372), so plots usually
    take the top of the weight-sorted list."""
    from .gaussian_process import inferenceObject
    return inferenceObject(obj).append({"Samples": list(obj["Samples"])[:n_samples]})

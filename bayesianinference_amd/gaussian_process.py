"""Host-side mirror of the reference's GP interface for the MI355X path.

The reference is Wolfram Language (no Wolfram kernel exists here or on the GPU box), so the host
side above the C ABI is Python, keeping the reference's names, argument meaning, association
keys and failure behaviour:

  inferenceObject                       BayesianUtilities.wl:107-138
  dataNormalForm                        BayesianUtilities.wl:203-221
  defineGaussianProcess                 BayesianGaussianProcess.wl:228-330  (BGP)
  defineInferenceProblem (GP subset)    BayesianStatistics.wl:164-308       (BS)
  predictFromGaussianProcess            BayesianGaussianProcess.wl:332-394

Differences forced by the closed kernel set (SURVEY.md §7 "Arbitrary WL kernels vs named
kernels"): `kernel` is a named spec ("SE", "SEARD", "Matern52", "Matern52ARD" or None for the
null kernel BGP:25).  The nugget and the mean function may be ANY functions of the point, as in
the reference (`nugget[points[[i]]]` BGP:37, `meanFunction /@ inputData` BGP:300): pass a callable
f(X[N, d], theta[p]) -> values[N] (vectorised over the points; the hyper-parameters it uses are
entries of theta) and the host evaluates it per theta and hands the values to the library
(gphip_*_pw); "Constant" / None keep the forms sigma_n^2 and mu / 0 read from theta.
`variables` must list the hyper-parameters in the C-ABI order (l.., sigma_f, sigma_n[, mu]).
All arithmetic runs in the HIP library; nothing here falls back to the CPU.
"""
from __future__ import annotations

import math
from collections.abc import Mapping

import numpy as np

from . import _lib

# BU:47 -- closed-source constant in the reference; the WL shim reads the real one at load time.
MACHINE_LOG_ZERO = -1.7976931348623157e308

_KERNEL_ALIASES = {
    "se": "se", "squaredexponential": "se", "se_iso": "se",
    "seard": "se_ard", "se_ard": "se_ard",
    "matern52": "matern52", "matern5/2": "matern52",
    "matern52ard": "matern52_ard", "matern52_ard": "matern52_ard",
    "null": "null", "none": "null",
    "matern32": "matern32", "matern3/2": "matern32", "matern32ard": "matern32_ard", "matern32_ard": "matern32_ard",
    "rq": "rq", "rationalquadratic": "rq", "rqard": "rq_ard", "rq_ard": "rq_ard",
}

# The WL expressions that define each named kernel for the *reference* defineGaussianProcess, so
# that "same inputs" is well defined.
WL_KERNEL_EXPRESSIONS = {
    "se": "Function[{p, q}, sf^2 Exp[-Total[(p - q)^2]/(2 l^2)]]",
    "se_ard": "Function[{p, q}, sf^2 Exp[-1/2 Total[((p - q)/{l1, ..., ld})^2]]]",
    "matern52": "Function[{p, q}, With[{s = Sqrt[Total[(p - q)^2]]/l}, sf^2 (1 + Sqrt[5] s + 5 s^2/3) Exp[-Sqrt[5] s]]]",
    "matern52_ard": "Function[{p, q}, With[{s = Sqrt[Total[((p - q)/{l1, ..., ld})^2]]}, sf^2 (1 + Sqrt[5] s + 5 s^2/3) Exp[-Sqrt[5] s]]]",
    "null": "Function[0]",
    "matern32": "Function[{p, q}, With[{s = Sqrt[Total[(p - q)^2]]/l}, sf^2 (1 + Sqrt[3] s) Exp[-Sqrt[3] s]]]",
    "matern32_ard": "Function[{p, q}, With[{s = Sqrt[Total[((p - q)/{l1, ..., ld})^2]]}, sf^2 (1 + Sqrt[3] s) Exp[-Sqrt[3] s]]]",
    "rq": "Function[{p, q}, sf^2 (1 + Total[(p - q)^2]/(2 alpha l^2))^-alpha]",
    "rq_ard": "Function[{p, q}, sf^2 (1 + Total[((p - q)/{l1, ..., ld})^2]/(2 alpha))^-alpha]",
}


def wl_kernel_expression(kname: str) -> str:
    """WL text of a named or composed kernel ('term [(+|*) term] [+const]'): what a user hands to the REFERENCE's
    defineGaussianProcess so that "same inputs" is defined (each term with its own l / alpha / sf symbols)."""
    if kname in WL_KERNEL_EXPRESSIONS:
        return WL_KERNEL_EXPRESSIONS[kname]
    key, offset = (kname[:-len("+const")], True) if kname.endswith("+const") else (kname, False)
    for op in ("+", "*"):
        if op in key:
            a, b = key.split(op, 1)
            body = f"({WL_KERNEL_EXPRESSIONS[a]})[p, q] {op} ({WL_KERNEL_EXPRESSIONS[b]})[p, q]"
            break
    else:
        body = f"({WL_KERNEL_EXPRESSIONS[key]})[p, q]"
    return "Function[{p, q}, " + ("c + " if offset else "") + body + "]"
WL_NUGGET_EXPRESSION = "Function[sn^2]"


class inferenceObject(Mapping):
    """Association wrapper (BU:107-138): obj[key], obj["Properties"], Normal -> .normal(),
    Append -> .append(), inferenceObject[$Failed] -> inferenceObject(None) with .failed."""

    def __init__(self, assoc):
        if isinstance(assoc, inferenceObject):          # BU:128 idempotent wrapping
            assoc = assoc._assoc
        self._assoc = None if assoc is None else dict(assoc)

    @property
    def failed(self) -> bool:                           # BU:126 FailureQ
        return self._assoc is None

    def normal(self) -> dict:                           # BU:121 Normal
        return dict(self._assoc or {})

    def append(self, extra: Mapping) -> "inferenceObject":   # BU:122-125 Append
        out = self.normal()
        out.update(extra)
        return inferenceObject(out)

    def __getitem__(self, key):
        if self._assoc is None:
            raise KeyError("inferenceObject[$Failed]")
        if key == "Properties":                         # BU:130
            return sorted(list(self._assoc) + ["Properties"])
        if isinstance(key, tuple):                      # obj["GaussianProcessData", "ModelFunctions"]
            cur = self._assoc
            for k in key:
                cur = cur[k]
            return cur
        return self._assoc[key]

    def __iter__(self):
        return iter(self._assoc or {})

    def __len__(self):
        return len(self._assoc or {})

    def __repr__(self):
        if self.failed:
            return "inferenceObject[$Failed]"
        return f"inferenceObject[<{len(self._assoc)} defined properties: {', '.join(list(self._assoc)[:4])}...>]"


def dataNormalForm(data):
    """BU:203-221: vector -> N x 1 matrix; (in, out) / {"Input","Output"} / [(x, y), ...] ->
    (X[N,d], Y[N,k]); None on malformed input ($Failed)."""
    try:
        if isinstance(data, Mapping) and "Input" in data:
            data = (data["Input"], data["Output"])
        if isinstance(data, tuple) and len(data) == 2:
            xin, xout = dataNormalForm(data[0]), dataNormalForm(data[1])
            if xin is None or xout is None or len(xin) != len(xout):
                return None
            return xin, xout
        if isinstance(data, list) and data and isinstance(data[0], tuple) and len(data[0]) == 2:
            return dataNormalForm(([p[0] for p in data], [p[1] for p in data]))   # {x -> y ..}
        arr = np.asarray(data, dtype=np.float64)
        if arr.ndim == 1:
            return arr[:, None]
        if arr.ndim == 2:
            return arr
    except (TypeError, ValueError):
        pass
    return None


def normalizeData(data_in, data_out=None):
    """BU:232-265: per-column standardisation (FeatureExtraction[.., "StandardizedVector"]: subtract the
    mean, divide by the sample standard deviation).  normalizeData(X) -> {"NormalizedData", "Function",
    "InverseFunction"}; normalizeData(X, Y) -> {"Input": {..}, "Output": {..}} (BU:239-242)."""
    if data_out is not None:
        return {"Input": normalizeData(data_in), "Output": normalizeData(data_out)}
    arr = dataNormalForm(data_in)
    if arr is None or isinstance(arr, tuple):
        return None
    mean = arr.mean(axis=0)
    sd = arr.std(axis=0, ddof=1) if len(arr) > 1 else np.ones(arr.shape[1])
    sd = np.where(sd > 0, sd, 1.0)

    def forward(x):
        return (dataNormalForm(x) - mean) / sd

    def inverse(z):
        return dataNormalForm(z) * sd + mean

    return {"NormalizedData": forward(arr), "Function": forward, "InverseFunction": inverse,
            "Mean": mean, "StandardDeviation": sd}


def normalizedDataQ(data) -> bool:
    """BU:267-286."""
    def test(a):
        return isinstance(a, Mapping) and {"NormalizedData", "Function", "InverseFunction"} <= set(a)
    return test(data) or (isinstance(data, Mapping) and set(data) == {"Input", "Output"}
                          and all(test(v) for v in data.values()))


def takePosteriorFraction(obj, frac: float):
    """BU:288-316: keep the heaviest samples until their cumulative CrudePosteriorWeight exceeds `frac`
    (frac = 1: all samples, sorted by decreasing weight)."""
    samples = sorted(obj["Samples"], key=lambda smp: -smp["CrudeLogPosteriorWeight"])
    if frac < 1:
        kept, count = [], 0.0
        for smp in samples:
            if count > frac:
                break
            kept.append(smp)
            count += smp["CrudePosteriorWeight"]
        samples = kept
    return inferenceObject(obj).append({"Samples": samples})


def _resolve_kernel(kernel):
    if kernel is None:
        return "null"
    if isinstance(kernel, _lib.CustomKernel):                     # ANY covariance function, as source text (gphip_create_custom)
        return kernel
    key = str(kernel).lower().replace(" ", "").replace("-", "")
    if key in _KERNEL_ALIASES:
        return _KERNEL_ALIASES[key]
    # composed forms: term [(+|*) term] [+const]  (include/gphip.h GPHIP_KERNEL_COMPOSE)
    body, offset = (key[:-len("+const")], "+const") if key.endswith("+const") else (key, "")
    for op in ("+", "*"):
        if op in body:
            a, b = body.split(op, 1)
            if a in _KERNEL_ALIASES and b in _KERNEL_ALIASES and "null" not in (_KERNEL_ALIASES[a], _KERNEL_ALIASES[b]):
                return _KERNEL_ALIASES[a] + op + _KERNEL_ALIASES[b] + offset
            break
    else:
        if body in _KERNEL_ALIASES and _KERNEL_ALIASES[body] != "null" and offset:
            return _KERNEL_ALIASES[body] + offset
    raise ValueError(f"kernel {kernel!r} is not a named kernel {sorted(set(_KERNEL_ALIASES.values()))} or a composed form "
                     "'term [(+|*) term] [+const]'; hand any other covariance function over as a _lib.CustomKernel (the source "
                     "text of its body, compiled at run time into the device kernel build)")


def _log_prior_function(prior, params):
    """'LogPriorPDFFunction' (BS:256-274): callable theta -> log pdf.  Accepts a callable, a list
    of scipy.stats frozen distributions (ProductDistribution), or "Uniform"/None over the box."""
    lo = np.array([p[1] for p in params], dtype=np.float64)
    hi = np.array([p[2] for p in params], dtype=np.float64)
    if callable(prior):
        return prior
    if prior is None or (isinstance(prior, str) and prior.lower() == "uniform"):
        logvol = float(np.sum(np.log(hi - lo)))

        def uniform_logpdf(theta):
            theta = np.asarray(theta, dtype=np.float64)
            return -logvol if np.all((theta >= lo) & (theta <= hi)) else MACHINE_LOG_ZERO
        return uniform_logpdf
    dists = list(prior)
    if len(dists) != len(params):
        raise ValueError("prior list length differs from the number of parameters")

    def product_logpdf(theta):
        theta = np.asarray(theta, dtype=np.float64)
        if not np.all((theta >= lo) & (theta <= hi)):
            return MACHINE_LOG_ZERO
        val = float(sum(d.logpdf(t) for d, t in zip(dists, theta)))
        return val if math.isfinite(val) else MACHINE_LOG_ZERO
    return product_logpdf


def random_domain_points(params, count=100, width=100.0, rng=None):
    """BU:366-372 randomDomainPointDistribution: Cauchy(0, width) truncated to the parameter box."""
    rng = rng or np.random.default_rng(0)
    lo = np.array([p[1] for p in params], dtype=np.float64)
    hi = np.array([p[2] for p in params], dtype=np.float64)
    u = rng.random((count, len(params)))
    clo = np.arctan(np.maximum(lo, -1e300) / width)
    chi = np.arctan(np.minimum(hi, 1e300) / width)
    return width * np.tan(clo + u * (chi - clo))


def _pointwise(fn, P, thetas):
    """values of a point-dependent nugget / mean function for every theta: [B, len(P)] (None stays None)"""
    if fn is None:
        return None
    rows = []
    for th in np.atleast_2d(thetas):
        v = np.asarray(fn(P, th), dtype=np.float64)
        rows.append(np.broadcast_to(v, (len(P),)) if v.ndim == 0 else v.reshape(len(P)))
    return np.ascontiguousarray(np.array(rows))


def make_log_likelihood(handle: "_lib.Handle", nugget_fn=None, mean_fn=None, X=None):
    """The drop-in closure for "LogLikelihoodFunction" (seam at BGP:249,293-294): theta -> machine
    real, total over the parameter box, $MachineLogZero on numerical failure (BGP:298-304), never an
    exception for bad theta values (BS:276-298).  nugget_fn / mean_fn: point-dependent nugget[x] / meanFunction[x]
    (callables f(X, theta) -> values[N], BGP:37, 300), evaluated here on the host for every theta."""
    pw = nugget_fn is not None or mean_fn is not None

    def log_likelihood(theta):
        theta = np.asarray(theta, dtype=np.float64)
        if theta.ndim == 2 or pw:                         # Listable use: B x p -> B (BGP:59)
            if pw:
                out, info = handle.loglik_batch_pw(theta, _pointwise(mean_fn, X, theta), _pointwise(nugget_fn, X, theta))
            else:
                out, info = handle.loglik_batch(theta)
            res = np.where((info == 0) & np.isfinite(out), np.clip(out, MACHINE_LOG_ZERO, -MACHINE_LOG_ZERO), MACHINE_LOG_ZERO)
            return res if theta.ndim == 2 else float(res[0])
        ll, info = handle.loglik(theta)
        if info != 0 or not math.isfinite(ll):
            return MACHINE_LOG_ZERO
        return min(max(ll, MACHINE_LOG_ZERO), -MACHINE_LOG_ZERO)     # Clip, BGP:183,190-197
    return log_likelihood


def defineGaussianProcess(data, kernel, nugget="Constant", meanFunction=None, variables=(),
                          variablePrior="Uniform", **rules) -> inferenceObject:
    """BGP:228-330.  data: (X, Y) with Y N x 1 (BGP:220-226); variables: [(name, min, max), ...] in
    the order (l.., sigma_f, sigma_n[, mu]); variablePrior: callable / list of scipy frozen
    distributions / "Uniform"; extra rules are forwarded into the object (`rest___Rule`, BGP:233,323)
    -- e.g. Device=0.  A caller-supplied LogLikelihoodFunction=callable is installed verbatim
    (BGP:293-294).  Returns inferenceObject(None) where the reference returns inferenceObject[$Failed]."""
    if normalizedDataQ(data) and isinstance(data, Mapping) and "Input" in data:      # BGP:214-218
        rules.setdefault("DataPreProcessors", {k: {"Function": v["Function"], "InverseFunction": v["InverseFunction"]}
                                               for k, v in data.items()})
        data = (data["Input"]["NormalizedData"], data["Output"]["NormalizedData"])
    norm = dataNormalForm(data)
    if norm is None or not isinstance(norm, tuple):
        return inferenceObject(None)                              # BGP:204-207 dataFormat
    X, Y = norm
    if Y.shape[1] != 1:                                           # BGP:220-226 outputDim
        return inferenceObject(None)
    if len(X) != len(Y):                                          # BGP:251-253
        return inferenceObject(None)
    if isinstance(nugget, str) and nugget.lower() != "constant":
        raise ValueError('nugget must be "Constant" (Function[sn^2]) or a callable f(X, theta) -> variances[N]')
    nugget_fn = nugget if callable(nugget) else None
    mean_fn = meanFunction if callable(meanFunction) else None
    kname = _resolve_kernel(kernel)
    mean = "zero" if (mean_fn is not None or meanFunction in (None, 0, "Zero", "zero")) else "const"
    if mean == "const" and str(meanFunction).lower() not in ("constant", "const"):
        raise ValueError("meanFunction must be None/0, 'Constant' or a callable f(X, theta) -> means[N]")
    params = [tuple(v) for v in variables]
    if not params or any(len(v) != 3 for v in params):            # paramSpecPattern, BS:19
        return inferenceObject(None)
    # Device = one ordinal; Devices = a list of ordinals -> ONE multi-device handle (the library shards a large
    # factorisation over them and deals batches / posterior samples / test points to them, include/gphip.h);
    # Precision = "Double" (fp64, the parity path) | "Single" (fp32 device arithmetic, BASELINE.json cfg 5)
    device = rules.pop("Devices", rules.pop("Device", None))
    precision = str(rules.pop("Precision", "Double")).lower()
    if precision not in ("double", "single"):
        raise ValueError('Precision must be "Double" or "Single"')
    handle = _lib.Handle(X, Y[:, 0], kname, mean, dtype=64 if precision == "double" else 32,
                         device=device)                           # raises loudly without the library / GPU
    if handle.p != len(params):
        handle.close()
        raise ValueError(f"kernel {kname!r} with mean {mean!r} on d={X.shape[1]} needs {handle.p} "
                         f"hyper-parameters (l.., sigma_f, sigma_n[, mu]); got {len(params)}")
    # Switch[logLikelihood, Automatic, .., _Function | _CompiledFunction, .., _, ..]  (BGP:272-307):
    #   callable      installed verbatim (BGP:293-294)
    #   "Automatic"   LogLikelihood[MultinormalDistribution[m, K], {y}], unevaluated -> $MachineLogZero (BGP:273-292):
    #                 the multivariate-normal log-pdf IS -1/2 (N log 2pi + log det K + r.K^-1 r), and "stays
    #                 unevaluated" = K not positive definite = info != 0 -- so on this path both branches are the
    #                 same device computation; the branch taken is recorded under "LikelihoodBranch"
    #   anything else the default closure (BGP:296-305)
    user_ll = rules.pop("LogLikelihoodFunction", None)
    branch = "UserFunction" if callable(user_ll) else ("Automatic" if isinstance(user_ll, str) and
                                                       user_ll.lower() == "automatic" else "Default")
    pointwise = nugget_fn is not None or mean_fn is not None
    loglik = user_ll if callable(user_ll) else make_log_likelihood(handle, nugget_fn, mean_fn, X)

    def log_likelihood_gradient(theta):
        """(value, gradient) -- extension for gradient-based MAP / Laplace routes (LA:177-238 uses
        NMaximize without gradients); sentinel and NaN gradient on numerical failure.  Defined for the constant nugget /
        mean forms only (the library cannot differentiate a host function of the point)."""
        if pointwise:
            return loglik(theta), np.full(len(params), np.nan)
        ll, grad, info = handle.loglik_grad(theta)
        return (ll, grad) if info == 0 else (MACHINE_LOG_ZERO, np.full(len(params), np.nan))

    def covariance_function(theta):                               # "CovarianceFunction", BGP:264-271
        K = handle.covariance(theta)                              # Listable: B x p -> B matrices (BGP:59)
        if nugget_fn is not None:                                 # nugget[points[[i]]] on the diagonal instead of sn^2
            th2 = np.atleast_2d(np.asarray(theta, dtype=np.float64))
            nug = _pointwise(nugget_fn, X, th2)
            Kb = K.reshape((len(th2),) + K.shape[-2:]).copy()
            for b in range(len(th2)):
                sn = th2[b, handle.p - (2 if mean == "const" else 1)]      # theta = (l.., sf, sn[, mu])
                Kb[b][np.diag_indices(len(X))] += nug[b] - sn * sn
            K = Kb.reshape(K.shape)
        if isinstance(kname, str) and kname == "null":            # covarianceMatrix[.., nullKernel, ..] = nugget /@ points:
            return np.diagonal(K, axis1=-2, axis2=-1).copy()      # the DIAGONAL as a vector (BGP:27)
        return K

    def inverse_covariance_function(theta):                       # "InverseCovarianceFunction", BGP:308
        if pointwise:
            nug, mt = _pointwise(nugget_fn, X, theta), _pointwise(mean_fn, X, theta)
            info = handle.fit_pw(theta, mt, nug)
        else:
            info = handle.fit(theta)
        if info != 0:
            return MACHINE_LOG_ZERO                               # Throw[$MachineLogZero, "MatInv"]
        return {"Inverse": handle.solve, "LogDet": handle.logdet()}

    return defineInferenceProblem({
        "Data": (X, Y),
        "PriorDistribution": variablePrior,
        "Parameters": params,
        "KernelName": kname, "MeanName": mean, "LikelihoodBranch": branch,
        "GaussianProcessData": {
            "ModelFunctions": {
                "KernelFunction": ((kname.name, kname.body) if isinstance(kname, _lib.CustomKernel)
                                   else (kname, wl_kernel_expression(kname))),
                "NuggetFunction": nugget_fn if nugget_fn is not None else WL_NUGGET_EXPRESSION,
                "MeanFunction": mean_fn if mean_fn is not None else mean,
                "CovarianceFunction": covariance_function,
                "InverseCovarianceFunction": inverse_covariance_function,
            },
            "HIPHandle": handle,
        },
        **rules,
        "LogLikelihoodGradientFunction": log_likelihood_gradient,
        "LogLikelihoodFunction": loglik,
    })


def defineInferenceProblem(assoc: Mapping) -> inferenceObject:
    """The subset of BS:164-308 a GP object goes through: parameter normal form, ParameterSymbols
    (BS:222), LogPriorPDFFunction (BS:256-274) and the 100-random-theta smoke test of both closures
    (BS:276-298) -- a closure that returns anything but finite reals fails the definition."""
    assoc = dict(assoc)
    params = assoc.get("Parameters")
    if not params or "LogLikelihoodFunction" not in assoc:
        return inferenceObject(None)
    assoc["ParameterSymbols"] = [p[0] for p in params]
    if "LogPriorPDFFunction" not in assoc:
        assoc["LogPriorPDFFunction"] = _log_prior_function(assoc.get("PriorDistribution"), params)
    pts = random_domain_points(params, 100)
    prior_vals = np.array([assoc["LogPriorPDFFunction"](t) for t in pts], dtype=np.float64)
    if not np.all(np.isfinite(prior_vals)):
        return inferenceObject(None)
    ll = assoc["LogLikelihoodFunction"]
    try:
        vals = np.asarray(ll(pts), dtype=np.float64)             # one batched sweep (B x p -> B)
        if vals.shape != (len(pts),):
            raise ValueError
    except (TypeError, ValueError):
        vals = np.array([ll(t) for t in pts], dtype=np.float64)
    if not np.all(np.isfinite(vals)):
        return inferenceObject(None)
    return inferenceObject(assoc)


def predictFromGaussianProcess(obj_or_examples, pts, kernel=None, theta=None, meanFunction=None):
    """BGP:332-394.  Two forms:
      predictFromGaussianProcess(obj, pts)            obj has "GaussianProcessData" and "Samples"
          (BGP:343-376): one Normal per posterior sample, mixed with the CrudePosteriorWeights.
          pts may be an int > 1: regular grid over the data bounds (BGP:332-341).
      predictFromGaussianProcess((X, Y), pts, kernel, theta[, meanFunction])
          direct form with the hyper-parameters baked in (BGP:378-394); duplicates in pts are
          removed first (DeleteDuplicates, BGP:380).
    Returns a dict with "Points" [M,d], "Weights" [S], "Mean" [S,M], "StandardDeviation" [S,M]:
    the content of Association[x* -> MixtureDistribution[weights, {NormalDistribution[mu, sigma]..}]]
    (for the direct form S = 1).  The variance includes the test-point nugget (BGP:113)."""
    if isinstance(obj_or_examples, inferenceObject):
        obj = obj_or_examples
        if obj.failed or "GaussianProcessData" not in obj or "Samples" not in obj:
            return None
        X = obj["Data"][0]
        if isinstance(pts, (int, np.integer)):
            if pts <= 1 or X.shape[1] != 1:
                return None
            pts = np.linspace(X.min(), X.max(), int(pts))         # CoordinateBoundsArray, BGP:336-339
        P = dataNormalForm(pts)
        if P is None or isinstance(P, tuple):
            return None
        handle = obj["GaussianProcessData"]["HIPHandle"]
        samples = obj["Samples"]
        points = np.array([s["Point"] for s in samples], dtype=np.float64)
        weights = np.array([s["CrudePosteriorWeight"] for s in samples], dtype=np.float64)
        mf = obj["GaussianProcessData"]["ModelFunctions"]
        nugget_fn = mf["NuggetFunction"] if callable(mf["NuggetFunction"]) else None
        mean_fn = mf["MeanFunction"] if callable(mf["MeanFunction"]) else None
        return _predict_samples(handle, points, weights, P, nugget_fn, mean_fn, X)
    norm = dataNormalForm(obj_or_examples)
    P = dataNormalForm(pts)
    if norm is None or not isinstance(norm, tuple) or P is None or norm[1].shape[1] != 1:
        return None                                               # Return[$Failed], BGP:383-390
    _, first = np.unique(P, axis=0, return_index=True)
    P = P[np.sort(first)]
    mean = "zero" if meanFunction in (None, 0, "Zero", "zero") else "const"
    handle = _lib.Handle(norm[0], norm[1][:, 0], _resolve_kernel(kernel), mean)
    try:
        return _predict_samples(handle, np.atleast_2d(np.asarray(theta, dtype=np.float64)), np.ones(1), P)
    finally:
        handle.close()


def predictiveDistribution(obj, inputs=None, estimate=None):
    """BS:1373-1416 for GP objects.  The reference's predictiveDistribution needs a "GeneratingDistribution"
    (BS:1381-1387), which a GP object does not carry; BASELINE.json's north_star names predictiveDistribution as the
    prediction API, so for objects with "GaussianProcessData" it forwards to predictFromGaussianProcess.
    estimate = "MaximumLikelihood" / "MAP" first reduces "Samples" to the single best sample (BS:1389-1416:
    TakeLargestBy LogLikelihood resp. LogLikelihood + LogPriorPDF).  Returns None ($Failed) for an unsampled object
    (BS:1375-1380) or a missing `inputs`."""
    if not isinstance(obj, inferenceObject) or obj.failed or "Samples" not in obj:
        return None                                               # predictiveDistribution::unsampled
    if "GaussianProcessData" not in obj or inputs is None:
        return None                                               # predictiveDistribution::MissGenDist
    if estimate is not None:
        key = {"maximumlikelihood": lambda smp: smp["LogLikelihood"],
               "map": lambda smp: smp["LogLikelihood"] + smp.get("LogPriorPDF", 0.0)}.get(str(estimate).lower())
        if key is None:
            raise ValueError('estimate must be None, "MaximumLikelihood" or "MAP"')
        obj = obj.append({"Samples": [max(obj["Samples"], key=key)]})
    return predictFromGaussianProcess(obj, inputs)


def _predict_samples(handle, points, weights, P, nugget_fn=None, mean_fn=None, X=None):
    """One batched pass over all samples (gphip_predict_samples); singular samples -> NaN rows.  Point-dependent
    nugget / mean functions are evaluated per sample at the training AND the test points (BGP:113, 408)."""
    if nugget_fn is not None or mean_fn is not None:
        mean, var, info = handle.predict_samples_pw(points, P, _pointwise(mean_fn, X, points), _pointwise(nugget_fn, X, points),
                                                    _pointwise(mean_fn, P, points), _pointwise(nugget_fn, P, points))
    else:
        mean, var, info = handle.predict_samples(points, P)
    bad = info != 0
    mean[bad], var[bad] = np.nan, np.nan
    with np.errstate(invalid="ignore"):
        sd = np.sqrt(var)                                          # Sqrt, BGP:414
    return {"Points": P, "Weights": weights, "Mean": mean, "StandardDeviation": sd}


def mixture_moments(pred: Mapping):
    """Mean and variance of the per-point MixtureDistribution (what regressionPlot1D draws, BV:310-374)."""
    w = np.asarray(pred["Weights"], dtype=np.float64)
    w = w / w.sum()
    mu, sd = pred["Mean"], pred["StandardDeviation"]
    m = w @ mu
    return m, w @ (sd ** 2 + mu ** 2) - m ** 2


def mixture_percentiles(pred: Mapping, levels: Sequence[float] = (0.95, 0.5, 0.05)):
    """InverseCDF of every per-point MixtureDistribution at `levels` -- the curves regressionPlot1D draws
    by default ("DistributionPercentiles" -> {0.95, 0.5, 0.05}, BV:303-308, 351-352).  Returns an array
    len(levels) x M.  The mixture CDF is monotone, so a bracketed bisection + Newton polish per point
    is exact to the last bits; everything is vectorised over the M test points."""
    from scipy.special import ndtr
    levels = np.asarray(levels, dtype=np.float64)
    if levels.ndim != 1 or levels.size == 0 or levels.min() <= 0.0 or levels.max() >= 1.0:
        raise ValueError("levels must lie strictly between 0 and 1 (BV:351)")
    w = np.asarray(pred["Weights"], dtype=np.float64)
    w = w / w.sum()
    mu = np.asarray(pred["Mean"], dtype=np.float64)
    sd = np.asarray(pred["StandardDeviation"], dtype=np.float64)
    keep = w > 0.0
    w, mu, sd = w[keep], mu[keep], sd[keep]

    def cdf(x):                                            # x: (M,) -> (M,)
        return np.einsum("s,sm->m", w, ndtr((x[None, :] - mu) / sd))

    def pdf(x):
        z = (x[None, :] - mu) / sd
        return np.einsum("s,sm->m", w, np.exp(-0.5 * z * z) / (sd * np.sqrt(2.0 * np.pi)))

    out = np.empty((levels.size, mu.shape[1]))
    span = 9.0 * sd                                       # every component's mass to ~1e-19 lies inside
    for k, q in enumerate(levels):
        lo, hi = (mu - span).min(axis=0), (mu + span).max(axis=0)
        for _ in range(60):                               # bisection: brackets shrink by 2^-60
            mid = 0.5 * (lo + hi)
            below = cdf(mid) < q
            lo = np.where(below, mid, lo)
            hi = np.where(below, hi, mid)
        x = 0.5 * (lo + hi)
        for _ in range(2):                                # Newton polish inside the bracket
            step = (cdf(x) - q) / np.maximum(pdf(x), 1e-300)
            x = np.clip(x - step, lo, hi)
        out[k] = x
    return out


def mixture_plot_moments(pred: Mapping):
    """The "Moments" option of regressionPlot1D (BV:340-349): per point the three curves
    {mean + sd + m3^(1/3), mean, mean - sd + m3^(1/3)} with m3 the third central moment of the mixture
    and the real cube root (Surd)."""
    w = np.asarray(pred["Weights"], dtype=np.float64)
    w = w / w.sum()
    mu = np.asarray(pred["Mean"], dtype=np.float64)
    sd = np.asarray(pred["StandardDeviation"], dtype=np.float64)
    m = w @ mu
    var = w @ (sd ** 2 + mu ** 2) - m ** 2
    dm = mu - m
    m3 = w @ (dm ** 3 + 3.0 * dm * sd ** 2)               # third central moment of a normal mixture
    s = np.sqrt(np.maximum(var, 0.0))
    skew = np.cbrt(m3)
    return np.stack([m + s + skew, m, m - s + skew])


# ---------------------------------------------------------------------------------------------
# persistence: an inferenceObject is a plain Association the user may
# Put/Export (BU:125); the device state (X, y resident; L, z after a fit) is rebuildable from it.
# ---------------------------------------------------------------------------------------------
def save_gaussian_process(obj, path: str, theta=None):
    """Writes data, kernel/mean names, parameter specs, samples (if any) and an optional fitted theta."""
    X, Y = obj["Data"]
    kern = obj["KernelName"]
    custom = isinstance(kern, _lib.CustomKernel)
    payload = {"X": X, "Y": Y, "kernel": "custom:" + kern.name if custom else kern, "mean": obj["MeanName"],
               "param_names": np.array([p[0] for p in obj["Parameters"]]),
               "param_lo": np.array([p[1] for p in obj["Parameters"]], dtype=np.float64),
               "param_hi": np.array([p[2] for p in obj["Parameters"]], dtype=np.float64)}
    if custom:                                                    # the function travels as its source text
        payload["kernel_body"] = kern.body
        payload["kernel_nparams"] = kern.nparams
    if "Samples" in obj:
        payload["sample_points"] = np.array([smp["Point"] for smp in obj["Samples"]], dtype=np.float64)
        payload["sample_logw"] = np.array([smp["CrudeLogPosteriorWeight"] for smp in obj["Samples"]])
    if theta is not None:
        payload["theta_fit"] = np.asarray(theta, dtype=np.float64)
    np.savez_compressed(path, **payload)


def load_gaussian_process(path: str, variablePrior="Uniform", trust_kernel_source: bool = False, **rules):
    """Rebuilds the object (new device handle, data uploaded again); if a fitted theta was saved the
    handle is re-fitted so predict/solve work immediately.  Returns (object, theta_fit or None).
    A checkpoint of a run-time compiled covariance function carries that function's C++ SOURCE TEXT, which loading
    compiles and runs on the device: such a file is executable content, not data.  It is refused unless the caller says
    the file comes from a trusted source (trust_kernel_source=True)."""
    z = np.load(path, allow_pickle=False)
    if "kernel_body" in z and not trust_kernel_source:
        raise ValueError(f"{path} carries the source text of a covariance function (kernel_body): loading compiles and runs it. "
                         "Pass trust_kernel_source=True if the file comes from a trusted source.")
    params = [(str(n), float(a), float(b)) for n, a, b in zip(z["param_names"], z["param_lo"], z["param_hi"])]
    kernel = None if str(z["kernel"]) == "null" else str(z["kernel"])
    if "kernel_body" in z:
        kernel = _lib.CustomKernel(str(z["kernel_body"]), int(z["kernel_nparams"]), name=str(z["kernel"])[len("custom:"):])
    mean = "Constant" if str(z["mean"]) == "const" else None
    obj = defineGaussianProcess((z["X"], z["Y"]), kernel, "Constant", mean, params, variablePrior, **rules)
    if obj.failed:
        return obj, None
    extra = {}
    if "sample_points" in z:
        logw = z["sample_logw"]
        extra["Samples"] = [{"Point": pt, "CrudeLogPosteriorWeight": float(lw), "CrudePosteriorWeight": float(np.exp(lw))}
                            for pt, lw in zip(z["sample_points"], logw)]
    theta = z["theta_fit"] if "theta_fit" in z else None
    if theta is not None:
        obj["GaussianProcessData"]["HIPHandle"].fit(theta)
    return (obj.append(extra) if extra else obj), theta

"""Is the evidence of the tabulated-prior sampler unbiased when the LIBRARY draws the starting pool (gphip_tab_prior_sample)?
Same problem as tests/test_gpu_wl_shim.py::test_native_sampler_with_a_tabulated_normal_prior..: z-scores of log Z against
quadrature over many seeds, pool given by numpy (rejection sampling) vs drawn by the library.   python scripts/gpu_tab_pool_bias.py [seeds]"""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_wl_shim import Kernel, NO_ERROR
from bayesianinference_amd import nested_sampling as ns
from scipy.stats import norm
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
K = Kernel()
rng = np.random.default_rng(3)
n = 40
X = rng.random((n, 1))
y = 0.3 + 0.8 * rng.standard_normal(n)
box = np.array([[0.4, 2.0], [-1.0, 1.5]])
m0, s0 = 0.2, 0.5
mass = norm.cdf(box[1, 1], m0, s0) - norm.cdf(box[1, 0], m0, s0)
g = 1200
sn = box[0, 0] + (np.arange(g) + 0.5) * (box[0, 1] - box[0, 0]) / g
mu = box[1, 0] + (np.arange(g) + 0.5) * (box[1, 1] - box[1, 0]) / g
s1, s2 = y.sum(), (y * y).sum()
quad = (s2 - 2 * mu[None, :] * s1 + n * mu[None, :] ** 2) / sn[:, None] ** 2
ll = -0.5 * (n * math.log(2 * math.pi) + 2 * n * np.log(sn)[:, None] + quad)
lp = -math.log(box[0, 1] - box[0, 0]) + norm.logpdf(mu, m0, s0)[None, :] - math.log(mass)
want = ns.log_sum_exp((ll + lp).ravel()) + math.log((box[0, 1] - box[0, 0]) / g * (box[1, 1] - box[1, 0]) / g)
nodes = 513
tab = np.stack([np.full(nodes, -math.log(box[0, 1] - box[0, 0])), norm.logpdf(np.linspace(box[1, 0], box[1, 1], nodes), m0, s0) - math.log(mass)])
rc, hs = K.call("gphip_wl_create", [X, y, 4, 1, 64, np.array([0])], "int")
pool = 60
for given in (True, False):
    zs = []
    for seed in range(nseeds):
        r = np.random.default_rng(1000 + seed)
        start = np.empty((pool, 2))
        start[:, 0] = r.uniform(box[0, 0], box[0, 1], pool)
        k = 0
        while k < pool:
            v = r.normal(m0, s0)
            if box[1, 0] <= v <= box[1, 1]:
                start[k, 1] = v; k += 1
        opts = np.array([pool, 10000, 100, 25, 32, 0.01, 0.0, 1.0, float(seed)])
        rc, rows = K.call("gphip_wl_nested_sampling_tab", [hs, box, tab, opts, start if given else np.zeros(0)])
        assert rc == NO_ERROR
        res = {"Points": rows[:, :2], "LogLikelihood": rows[:, 2], "LogPriorPDF": rows[:, 3], "AcceptanceRate": rows[:, 4],
               "SamplePoolSize": pool, "GeneratedNestedSamples": len(rows) - pool, "TotalSamples": len(rows)}
        out = ns.evidence_sampling(res, ["sn", "mu"], pool, np.random.default_rng(seed))
        zs.append((out["LogEvidence"]["Mean"] - want) / out["LogEvidence"]["StandardError"])
    zs = np.array(zs)
    print(f"pool {'given' if given else 'drawn by the library'}: {nseeds} seeds, z mean {zs.mean():+.3f} +- {zs.std(ddof=1) / math.sqrt(nseeds):.3f}, std {zs.std(ddof=1):.3f}, "
          f"min {zs.min():.2f} max {zs.max():.2f}", flush=True)

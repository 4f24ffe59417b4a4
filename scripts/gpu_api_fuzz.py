"""Randomised sequences of C-ABI calls on one handle (single-device or virtual-rank multi-device), every numeric result
checked against numpy on the covariance matrix the LIBRARY itself returns (so no oracle import: this is a developer
script): likelihood parts, batch, fit -> predict / solve / logdet, gradient vs finite differences of the likelihood,
cross covariance, option flips in between.  Finds state-machine bugs (stale fitted flags, scratch buffers reused at a
different size, slots re-allocated under a resident factor)."""
import os
os.environ["GPHIP_TEST_HOOKS"] = "1"      # (the fault-injection option names exist only with this)
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t_end = time.time() + budget
ncalls = nh = 0
log = open(os.path.join(ROOT, 'gpurun_out', 'fuzz_ops.log'), 'w') if os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else open(os.devnull, 'w')
LOG2PI = np.log(2 * np.pi)

def ref_parts(K, r):
    L = np.linalg.cholesky(K)
    z = np.linalg.solve(L, r)
    ld, qd = 2 * np.log(np.diag(L)).sum(), float(z @ z)
    return -0.5 * (len(r) * LOG2PI + ld + qd), ld, qd

RT = 1e-8
def close(a, b, n, rtol=None):
    return abs(a - b) <= (RT if rtol is None else rtol) * max(abs(b), float(n))

while time.time() < t_end:
    n = int(rng.choice([1, 5, 64, 127, 128, 129, 300, 513, 777, 1100, 1537, 2500, 4200], p=[.06, .06, .08, .08, .08, .08, .1, .1, .1, .1, .08, .05, .03]))
    dtype = int(rng.choice([64, 64, 64, 32]))
    RT = 1e-8 if dtype == 64 else 2e-3                    # fp32 device arithmetic: 1e-3-class agreement
    d = int(rng.choice([1, 2, 3, 8]))
    kernel = str(rng.choice(["se", "se_ard", "matern52", "matern52_ard"]))     # (default_theta knows these four)
    mean = str(rng.choice(["zero", "const"]))
    world = int(rng.choice([1, 1, 2, 3, 4]))
    X, y = syn.make_dataset(n, d, seed=int(rng.integers(1 << 30)))
    as_text = kernel == "se_ard" and rng.random() < 0.4   # the same function handed over as source text (gphip_create_custom)
    kobj = _lib.CustomKernel("T s = 0; for (int k = 0; k < D; ++k) { const T u = (X(k) - Y(k)) / P(k); s += u * u; } "
                             "return P(D) * P(D) * exp((T)-0.5 * s);", d + 1) if as_text else kernel
    h = _lib.Handle(X, y, kobj, mean, dtype=dtype, device=([0] * world if world > 1 else None))
    if world > 1:
        h.set_option("shard_min_n", int(rng.choice([0, 1 << 30])))
    nh += 1
    print(f'HANDLE n={n} d={d} {kernel}{" (as source text)" if as_text else ""} {mean} world={world} dtype={dtype}', file=log, flush=True)
    def rand_theta():
        th = syn.default_theta(kernel, d) * (0.6 + 0.8 * rng.random(len(syn.default_theta(kernel, d))))
        th[-1] = 0.1 + 0.4 * rng.random()
        return np.append(th, rng.normal(0, 0.3)) if mean == "const" else th
    fitted = None
    opts = {}
    for _ in range(int(rng.integers(4, 14))):
        op = str(rng.choice(["parts", "batch", "fit", "predict", "solve", "grad", "cross", "option", "samples", "nasty"]))
        ncalls += 1
        print(f'  op {op}', file=log, flush=True)
        if op == "option":
            name = str(rng.choice(["panel", "dataflow", "lookahead", "thin_tiles", "fused_eval", "dataflow_fine_nt", "panel_wide",
                                   "dataflow_tail", "grad_potri", "max_slots", "trsv", "kbuild_mfma", "latency_gemm", "shard_min_n", "panel_left",
                                   "replicate_factor", "share_local_panels", "debug_fail_alloc", "supertile", "dataflow_lds_kib", "fuse_potrf", "bcast_chunks",
                                   "dist_panel_df", "dist_owner_yield", "bcast_two_hop", "panel_df", "kbuild_mfma", "kbuild_mfma_bound", "custom_grad"]))
            if name == "debug_fail_alloc" and rng.random() < 0.7:
                name = "panel"
            val = {"panel": int(rng.choice([1, 2, 3, 4, 6])), "dataflow_fine_nt": int(rng.choice([0, 96])),
                   "debug_fail_alloc": int(rng.choice([1, 3, 7, 10])),
                   "dataflow_tail": int(rng.choice([0, 7, 64])), "max_slots": int(rng.choice([1, 3, 256])),
                   "shard_min_n": int(rng.choice([0, 1 << 30])), "panel_left": int(rng.choice([-1, 0, 1])), "supertile": int(rng.choice([0, 2, 3])), "dataflow_lds_kib": int(rng.choice([-1, 0, 84])),
                   "dist_panel_df": int(rng.choice([-1, 0, 1, 2, 3])), "panel_df": int(rng.choice([-1, 0, 1])),
                   "kbuild_mfma": int(rng.choice([0, 1, 2])), "kbuild_mfma_bound": int(rng.choice([1, 64, 512]))}.get(name, int(rng.integers(0, 2)))
            print(f'    {name}={val}', file=log, flush=True)
            if name == "debug_fail_alloc":
                # fault injection: the next slot (re)allocation fails at its val-th device allocation; the call must
                # report it, the handle must come back from the zero-slot state with the next call
                h.set_option("max_slots", 256)
                h.set_option(name, val)
                big = np.stack([rand_theta() for _ in range(40)])
                try:
                    h.loglik_batch(big)                    # grows past every earlier slot count -> hits the injection
                except _lib.GphipError as exc:
                    assert "failed at" in str(exc), exc
                h.set_option(name, 0)
                out, info = h.loglik_batch(big[:3])
                assert np.all(info == 0) and np.all(np.isfinite(out))
                fitted = None
                continue
            h.set_option(name, val)
            opts[name] = val
            if name == "panel" and world > 1:
                fitted = None          # documented: a DISTRIBUTED factor is laid out for the panel width it was made with;
                                       # prediction after a change of "panel" is refused (GPHIP_ERR_STATE: fit again)
            continue
        if op == "nasty":
            # hyper-parameters the closure must survive (BS:276-298: total over the box, sentinel on failure): NaN, 0,
            # 1e-12 nuggets, huge / tiny scales -- never an exception, info in {0, 1, 2}, finite value whenever info = 0,
            # and the handle keeps working afterwards
            B = int(rng.integers(1, 10))
            Th = np.stack([rand_theta() for _ in range(B)])
            for b in range(B):
                k = int(rng.integers(Th.shape[1]))
                Th[b, k] = rng.choice([np.nan, 0.0, 1e-12, 1e-300, 1e300, -1.0, np.inf])
            out, info = h.loglik_batch(Th)
            assert set(info.tolist()) <= {0, 1, 2} and np.all(np.isfinite(out[info == 0])), (op, Th, out, info)
            for b in range(min(B, 2)):
                h.fit(Th[b]); h.loglik_grad(Th[b])
            fitted = None
            th = rand_theta()
            Kn = h.covariance(th)
            ll, ld, qd, inf = h.loglik_parts(th)
            assert inf == 0 and close(ll, ref_parts(Kn, y - (th[-1] if mean == "const" else 0.0))[0], n), (op, "after")
            continue
        th = rand_theta()
        mu0 = th[-1] if mean == "const" else 0.0
        if op in ("parts", "grad", "fit"):
            K = h.covariance(th); fitted = None
        if op == "parts":
            ll, ld, qd, info = h.loglik_parts(th)
            w = ref_parts(K, y - mu0)
            assert info == 0 and close(ll, w[0], n) and close(ld, w[1], n) and close(qd, w[2], n), (op, n, d, kernel, world)
            fitted = None
        elif op == "batch":
            B = int(rng.integers(1, 12))
            Th = np.stack([rand_theta() for _ in range(B)])
            out, info = h.loglik_batch(Th)
            for b in (0, B - 1):
                Kb = h.covariance(Th[b])
                wb = ref_parts(Kb, y - (Th[b][-1] if mean == "const" else 0.0))[0]
                assert info[b] == 0 and close(out[b], wb, n), (op, n, d, kernel, mean, world, B, b, info.tolist(), out[b], wb, opts)
            fitted = None
        elif op == "fit":
            assert h.fit(th) == 0
            fitted = (th, K)
        elif op == "grad":
            ll, g, info = h.loglik_grad(th)
            assert info == 0 and close(ll, ref_parts(K, y - mu0)[0], n)
            k = int(rng.integers(len(th)))
            e = np.zeros(len(th)); e[k] = 1e-5 * max(abs(th[k]), 0.1)
            fd = (h.loglik(th + e)[0] - h.loglik(th - e)[0]) / (2 * e[k])
            if dtype == 64:
                assert abs(g[k] - fd) <= 2e-4 * max(abs(fd), abs(g).max(), 1.0), (op, n, k, g[k], fd)
            fitted = None                                   # the finite differences overwrote the factor
        elif op in ("predict", "solve") and fitted is not None:
            thf, Kf = fitted
            muf = thf[-1] if mean == "const" else 0.0
            if op == "solve":
                nr = int(rng.choice([1, 3, 130]))
                Bm = rng.standard_normal((n, nr))
                got = h.solve(Bm[:, 0] if nr == 1 else Bm)
                want = np.linalg.solve(Kf, Bm)
                np.testing.assert_allclose(got.reshape(n, -1), want, rtol=10 * RT, atol=10 * RT * np.abs(want).max())
                assert close(h.logdet(), np.linalg.slogdet(Kf)[1], n)
            else:
                M = int(rng.choice([1, 7, 200, 1100]))
                Xs = syn.make_test_points(M, d, seed=int(rng.integers(1 << 30)))
                k, kappa = None, None
                mu, var = h.predict(Xs)
                k, kappa = h.cross_covariance(thf, Xs)     # (un-fits the handle: refit below if needed)
                alpha = np.linalg.solve(Kf, y - muf)
                np.testing.assert_allclose(mu, muf + k.T @ alpha, rtol=100 * RT, atol=100 * RT)
                np.testing.assert_allclose(var, kappa - np.sum(k * np.linalg.solve(Kf, k), axis=0), rtol=100 * RT, atol=100 * RT)
                fitted = None
        elif op == "cross":
            Xs = syn.make_test_points(int(rng.choice([1, 9, 300])), d)
            k, kappa = h.cross_covariance(th, Xs)
            Kj = h.covariance(th)
            assert k.shape == (n, len(Xs)) and np.all(np.isfinite(k)) and np.allclose(kappa, Kj[0, 0])
            fitted = None
        elif op == "samples":
            S = int(rng.integers(1, 6))
            Th = np.stack([rand_theta() for _ in range(S)])
            Xs = syn.make_test_points(int(rng.choice([1, 40])), d)
            mS, vS, iS = h.predict_samples(Th, Xs)
            s = int(rng.integers(S))
            assert h.fit(Th[s]) == 0
            m1, v1 = h.predict(Xs)
            np.testing.assert_allclose(mS[s], m1, rtol=10 * RT, atol=10 * RT)
            np.testing.assert_allclose(vS[s], v1, rtol=10 * RT, atol=10 * RT)
            fitted = (Th[s], h.covariance(Th[s])); fitted = None
    h.close()
print(f"api fuzz: {nh} handles, {ncalls} calls, 0 failures", flush=True)

"""bench.py -- GP log-marginal-likelihood evaluations per second (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 32768] [--d 8]

A "step" is one pass of the hot path over one batch of synthetic input: one evaluation of
theta -> log p(y | X, theta) (kernel-matrix build + Cholesky + log|K| + quadratic form) at
N=32768, d=8, SE-ARD, fp64, with X and y already resident in HBM (`gphip_create` ran before the
timed region).  Every step uses a different theta (nested sampling never repeats a point).

Multi-GPU (N>1, launched by torch.distributed.run, one rank per GPU): the path shards over theta
-- independent likelihood evaluations, exactly how the reference's callers consume the closure
(BS:902-916 sweep, BS:1349 replicas) -- so ranks evaluate disjoint theta with NO data-path
collective ("weak" scaling); torch.distributed (RCCL) is used only for the timing barrier and the
max-over-ranks reduction.

One JSON line on rank 0, with `roofline` (dominant kernel = trailing SYRK on fp64 MFMA, timed
with HIP events on the library's own stream, inside the timed region) and `cpu_baseline`
(the CPU oracle = LU restatement of the reference algorithm, bounded sample, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (v_mfma_f64_16x16x4_f64)


def cpu_baseline(n_full: int, d: int, sample_n: int = 4096, reps: int = 2) -> dict:
    """Times the CPU oracle (scipy LAPACK LU, the algorithm LinearSolve uses) on a bounded sample
    and scales it to the metric's unit (evals/s at n_full): the kernel-matrix build with its
    quadratic cost, the factorisation + solve with its cubic cost, timed separately.  For
    information it also times a Cholesky variant of the same sample and cfg 1 (N=512, d=1) whole."""
    from oracle import gp_oracle as orc
    from bayesianinference_amd import synthetic as syn
    import scipy.linalg as sla
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    X, y = syn.make_dataset(sample_n, d)
    th = syn.default_theta("se_ard", d)
    orc.log_likelihood("se_ard", th, X[:512], y[:512])

    def limited(nthreads):
        from contextlib import nullcontext
        try:
            from threadpoolctl import threadpool_limits
            return threadpool_limits(limits=nthreads)
        except Exception:
            return nullcontext()

    def timed(nthreads):
        """(build s, LU factor + solve + formula s) per evaluation, oracle functions only."""
        tb = tf = 0.0
        with limited(nthreads):
            for i in range(reps):
                thi = th * (1.0 + 0.01 * i)
                t0 = time.perf_counter()
                r = orc.residual("se_ard", thi, X, y)
                K = orc.covariance_matrix("se_ard", thi, X)
                t1 = time.perf_counter()
                solve, logdet = orc.matrix_inverse_and_det(K)
                orc.gp_log_likelihood_from_parts(r, solve, logdet)
                t2 = time.perf_counter()
                tb += t1 - t0
                tf += t2 - t1
        return tb / reps, tf / reps

    # LAPACK on very many threads can be slower than on fewer: report the best of a short ladder
    ladder = sorted({t for t in (16, 32, 64, threads) if t <= threads})
    results = {t: timed(t) for t in ladder}
    threads = min(results, key=lambda t: sum(results[t]))
    tb, tf = results[threads]
    s2, s3 = (n_full / sample_n) ** 2, (n_full / sample_n) ** 3
    est = tb * s2 + tf * s3

    with limited(threads):                       # information only: SPD-aware variant, and cfg 1 as is
        K = orc.covariance_matrix("se_ard", th, X)
        t0 = time.perf_counter()
        c = sla.cho_factor(K, lower=True, overwrite_a=True, check_finite=False)
        sla.cho_solve(c, y, check_finite=False)
        t_chol = time.perf_counter() - t0
        X1, y1 = syn.make_dataset(512, 1)
        th1 = syn.default_theta("se", 1)
        t0 = time.perf_counter()
        for _ in range(5):
            orc.log_likelihood("se", th1, X1, y1)
        t_cfg1 = (time.perf_counter() - t0) / 5
    return {"value": 1.0 / est, "unit": "evals/s", "cores": int(threads), "kind": "port",
            "sample": f"CPU oracle (numpy build + scipy dgetrf/dgetrs LU restatement of BGP:29-43,130-141,"
                      f"181-199; not Mathematica) timed at N={sample_n} d={d} x{reps}: build {tb:.3f} s "
                      f"scaled by (N/{sample_n})^2={s2:.0f}, LU+solve {tf:.3f} s scaled by "
                      f"(N/{sample_n})^3={s3:.0f} => {est:.1f} s/eval at N={n_full}",
            "also": {"cholesky_variant_s_at_sample": round(t_chol, 4),
                     "cfg1_N512_d1_evals_per_s": round(1.0 / t_cfg1, 2)}}


def pmc_traffic(kernel_substr: str = "gemm_nt_kernel<double, 0,"):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_summary.csv: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command,
    FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  None if no summary is present."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.csv"))):
        fetch = write = None
        with open(path) as f:
            for line in f:
                parts = line.strip().split(",")
                # kernel names contain a comma ("<double, 0>"): take the fields from the right
                if len(parts) >= 5 and kernel_substr in ",".join(parts[:-4]):
                    if parts[-4] == "FETCH_SIZE":
                        fetch = float(parts[-2])
                    elif parts[-4] == "WRITE_SIZE":
                        write = float(parts[-2])
        if fetch is not None and write is not None:
            best = {"bytes_per_launch": (2.0 * fetch + write) * 1024.0, "source": os.path.basename(path)}
    return best


def other_configs(local_rank: int) -> dict:
    """Short measurements of the other BASELINE.json configs on one GPU (reported next to the headline,
    never part of `value`): cfg 1 / cfg 2 = one theta at a time at N=512 d=1 and N=8192 d=8 (the
    reference's sequential-chain usage, latency bound), cfg 4 = 200 theta x N=4096 batched, cfg 5 =
    Matern-5/2 N=65536 d=16 fp32 fit + prediction on 10k test points."""
    from bayesianinference_amd import _lib, synthetic as syn
    out = {}
    try:
        for name, n, d, kernel, reps in (("cfg1_n512_d1_f64", 512, 1, "se", 200), ("cfg2_n8192_d8_f64", 8192, 8, "se_ard", 10)):
            X, y = syn.make_dataset(n, d)
            th = syn.default_theta(kernel, d)
            h = _lib.Handle(X, y, kernel, device=local_rank)
            h.loglik(th); h.loglik(th)
            t0 = time.perf_counter()
            for _ in range(reps):
                _, info = h.loglik(th)
            dt = (time.perf_counter() - t0) / reps
            out[name] = {"ms_per_eval": dt * 1e3, "evals_per_s": 1.0 / dt, "info": int(info)}
            h.close()
    except Exception as exc:                                    # never let an extra break the headline
        out["cfg1_cfg2_error"] = repr(exc)
    try:
        X, y = syn.make_dataset(4096, 8)
        Th = syn.theta_batch(200, "se_ard", 8)
        Th[:, -1] = np.maximum(Th[:, -1], 0.05)
        h = _lib.Handle(X, y, "se_ard", device=local_rank)
        h.loglik_batch(Th[:8])
        h.loglik_batch(Th)
        t0 = time.perf_counter()
        _, info = h.loglik_batch(Th)
        dt = time.perf_counter() - t0
        out["cfg4_batch_200x4096_f64"] = {"evals_per_s": 200 / dt, "tflops": 200 * 4096 ** 3 / 3 / dt / 1e12,
                                          "failed": int((info != 0).sum())}
        h.close()
    except Exception as exc:                                    # never let an extra break the headline
        out["cfg4_error"] = repr(exc)
    try:
        n, d, m = 65536, 16, 10000
        X, y = syn.make_dataset(n, d)
        th = syn.default_theta("matern52_ard", d, dtype="f32")
        h = _lib.Handle(X, y, "matern52_ard", dtype=32, device=local_rank)
        h.loglik(th)
        t0 = time.perf_counter()
        info = h.fit(th)
        tf = time.perf_counter() - t0
        t0 = time.perf_counter()
        mu, var = h.predict(syn.make_test_points(m, d))
        tp = time.perf_counter() - t0
        out["cfg5_matern52_n65536_d16_f32"] = {"fit_ms": tf * 1e3, "cholesky_tflops": n ** 3 / 3 / tf / 1e12,
                                               "predict_10k_ms": tp * 1e3, "info": int(info),
                                               "finite": bool(np.all(np.isfinite(mu)) and np.all(var > 0))}
        h.close()
    except Exception as exc:
        out["cfg5_error"] = repr(exc)
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--panel", type=int, default=0)
    ap.add_argument("--mode", choices=["theta", "cholesky"], default="theta",
                    help="N>1 only. theta (default): ranks evaluate disjoint theta, no data-path collective, "
                         "weak scaling.  cholesky: ONE evaluation per step sharded over all ranks with the 1-D "
                         "block-cyclic Cholesky (RCCL broadcast of factored panels), strong scaling.")
    ap.add_argument("--supertile", type=int, default=0, help="experiment: XCD-private 8x8 super-tile order")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the short BASELINE.json cfg-4 / cfg-5 measurements")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch
    from bayesianinference_amd import _lib, synthetic as syn

    if not torch.cuda.is_available() or _lib.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (gfx950): the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n, d = args.n, args.d
    X, y = syn.make_dataset(n, d)                      # every rank regenerates the same data
    h = _lib.Handle(X, y, "se_ard", device=local_rank)
    if args.panel:
        h.set_option("panel", args.panel)
    if args.supertile:
        h.set_option("supertile", 1)
    base = syn.default_theta("se_ard", d)
    total_steps = args.warmup + args.steps
    # disjoint theta per rank and step: jitter the length-scales by < 5 %
    jit = syn.uniform(syn.STREAM_THETA, 1000 + rank * total_steps * (d + 2), total_steps * (d + 2))
    thetas = base[None, :] * (1.0 + 0.05 * (jit.reshape(total_steps, d + 2) - 0.5))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sharded = dist is not None and args.mode == "cholesky"
    if sharded:
        # every rank steps through the SAME theta; one likelihood is factored by all GPUs together
        from bayesianinference_amd.dist_cholesky import DistributedCholesky, TorchDistComm
        jit0 = syn.uniform(syn.STREAM_THETA, 1000, total_steps * (d + 2))
        thetas = base[None, :] * (1.0 + 0.05 * (jit0.reshape(total_steps, d + 2) - 0.5))
        dc = DistributedCholesky({rank: h}, TorchDistComm(dist), device=local_rank)

        def evaluate(th):
            ll, _, _, info = dc.loglik(th)
            return ll, info
    else:
        evaluate = h.loglik

    for i in range(args.warmup):
        evaluate(thetas[i])
    h.set_option("profile", 1)                         # events around the trailing SYRK launches only
    h.reset_profile()
    barrier()
    t0 = time.perf_counter()
    vals = []
    for i in range(args.warmup, total_steps):
        ll, info = evaluate(thetas[i])
        vals.append((ll, info))
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = h.profile()
    bad = [v for v in vals if v[1] != 0 or not np.isfinite(v[0])]
    if bad:
        raise SystemExit(f"bench: evaluation failed: {bad[:3]}")

    if rank == 0:
        syrk = prof["syrk_trailing"]
        achieved = syrk["flops"] / (syrk["ms"] * 1e-3) / 1e12 if syrk["ms"] > 0 else 0.0
        evals = args.steps * (1 if sharded else world)
        chol_flops = n ** 3 / 3.0
        out = {
            "metric": "GP log-marg-lik evals/sec at N=32768, d=8; Cholesky TFLOP/s vs fp64 peak",
            "value": evals / dt, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"SE-ARD GP log marginal likelihood, N={n} d={d} fp64, one theta per "
                                   f"step per GPU (kernel build + Cholesky + log|K| + quad form)",
                       "N": n, "d": d, "kernel": "se_ard", "parallelism": (f"1-D block-cyclic Cholesky over {world} GPUs (RCCL panel broadcast)" if sharded
                                       else f"theta-sharded x{world}")},
            "cholesky_tflops_per_gpu": chol_flops * args.steps / dt / 1e12 / (world if sharded else 1),
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_kernel<double, 0, 2, 2, 2> (trailing SYRK, v_mfma_f64_16x16x4_f64)",
                         "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "launches": int(syrk["launches"]), "avg_launch_ms": syrk["ms"] / max(syrk["launches"], 1),
                         "traffic": None},
        }
        tr = pmc_traffic()
        if tr is not None:
            out["roofline"]["traffic"] = tr["bytes_per_launch"]
            out["roofline"]["traffic_unit"] = "HBM-side bytes per launch (rocprofv3 PMC, " + tr["source"] + ")"
            out["roofline"]["algorithmic_bytes_per_launch"] = syrk["bytes"] / max(syrk["launches"], 1)
        if world == 1 and not args.no_extras:
            h.close()                                           # free the 8.7 GB workspace first
            out["other_configs"] = other_configs(local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, d)
        print(json.dumps(out), flush=True)
    h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

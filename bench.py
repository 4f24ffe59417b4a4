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
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (v_mfma_f64_16x16x4_f64)
FP32_MFMA_PEAK_TFLOPS = 157.3     # v_mfma_f32_16x16x4_f32: 64 flop/clk/SIMD
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def _roof(cls: dict, peak: float, unit: str, what: str) -> dict:
    """roofline entry from one profile class (HIP events on the library's own stream): algorithmic flops or bytes
    per launch / average launch time"""
    if cls["ms"] <= 0 or cls["launches"] <= 0:
        return {}
    work = cls["flops"] if unit == "TFLOP/s" else cls["bytes"]
    ach = work / (cls["ms"] * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
    return {"kernel": what, "bound": "mfma" if unit == "TFLOP/s" else "hbm", "achieved": ach, "peak": peak, "unit": unit,
            "frac": ach / peak, "launches": int(cls["launches"]), "avg_launch_ms": cls["ms"] / cls["launches"],
            "algorithmic_per_launch": work / cls["launches"]}


def cpu_baseline(n_full: int, d: int, max_full_s: float = 600.0) -> dict:
    """Times the CPU oracle (numpy kernel-matrix build + scipy LAPACK LU, the algorithm LinearSolve uses) WHOLE, on the
    host cores of this box: one evaluation at N = 16384 (LAPACK thread-count ladder there) and -- unless the N = 16384
    sample predicts more than `max_full_s` seconds ("32768 if < 10 min") -- one at the metric's own
    size N = n_full.  `value` is then MEASURED, not extrapolated; the cubic-law prediction from the smaller samples is
    kept under `also` so the two can be compared.  For information it also times a Cholesky variant and cfg 1
    (N=512, d=1) whole."""
    from oracle import gp_oracle as orc
    from bayesianinference_amd import synthetic as syn
    import scipy.linalg as sla
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1

    def limited(nthreads):
        from contextlib import nullcontext
        try:
            from threadpoolctl import threadpool_limits
            return threadpool_limits(limits=nthreads)
        except Exception:
            return nullcontext()

    th = syn.default_theta("se_ard", d)
    Xw, yw = syn.make_dataset(512, d)
    orc.log_likelihood("se_ard", th, Xw, yw)                   # warm the BLAS threads
    # The kernel-matrix build of the oracle is numpy element-wise code: serial unless its row blocks go to a thread pool
    # (round 3 timed it serial: 23 of the 39 s at N = 32768).  Same arithmetic, bit-identical matrix.
    build_threads = max(1, min(threads, 32))

    def timed(n, nthreads):
        """(build s, LU factor + solve + formula s, log-likelihood) of ONE whole evaluation, oracle functions only (the
        same calls orc.log_likelihood makes, split so that the two phases can be reported separately)."""
        X, y = syn.make_dataset(n, d)
        orc.BUILD_THREADS = build_threads
        try:
            with limited(1):                                     # (the pool's workers are the parallelism of the build)
                t0 = time.perf_counter()
                r = orc.residual("se_ard", th, X, y)
                K = orc.covariance_matrix("se_ard", th, X)
                t1 = time.perf_counter()
        finally:
            orc.BUILD_THREADS = 1
        with limited(nthreads):
            big = K.size > orc._BLOCK_ELEMS                      # large N: LU in place (K is symmetric bit for bit)
            solve, logdet = orc.matrix_inverse_and_det(K.T if big else K, overwrite=big)
            del K
            ll = orc.gp_log_likelihood_from_parts(r, solve, logdet)
            t2 = time.perf_counter()
        return t1 - t0, t2 - t1, ll

    # LAPACK on very many threads can be slower than on fewer: pick the LU's thread count on a short ladder AT N = 16384 (round 3
    # chose it at N = 4096, round 4 at N = 8192 -- where the first LU of the process ran 1.5x slower than the N = 16384 one that
    # followed it: cold pages / thread pool).  Every size is therefore WARMED once (an untimed LU of that size) before it is
    # timed, and the ladder stops as soon as more threads have lost twice.
    n_lad = 16384 if n_full > 16384 else max(2048, n_full // 2)
    Xl, yl = syn.make_dataset(n_lad, d)
    orc.BUILD_THREADS = build_threads
    try:
        with limited(1):
            Kl = orc.covariance_matrix("se_ard", th, Xl)
    finally:
        orc.BUILD_THREADS = 1
    probe = {}
    worse = 0
    for i, t in enumerate(sorted({t for t in (8, 16, 32, 64, 128) if t <= threads} or {threads})):
        with limited(t):
            for rep in range(2 if i == 0 else 1):               # (first entry: one untimed LU warms the size)
                Kc = Kl.copy()
                t0 = time.perf_counter()
                sla.lu_factor(Kc, overwrite_a=True, check_finite=False)
                probe[t] = time.perf_counter() - t0
        worse = worse + 1 if probe[t] > min(probe.values()) else 0
        if worse >= 2:
            break
    del Kl, Kc
    threads = min(probe, key=probe.get)
    measured = {}
    for n in (16384,):
        if n < n_full:
            tb, tf, ll = timed(n, threads)                       # (the ladder above has warmed this size)
            measured[n] = (tb, tf, ll)
    # cfg 2's size for information: LAPACK's best thread count is size dependent (on the 256-CPU hosts of this pool the LU at
    # N = 8192 ran SLOWER on 16 threads than the N = 16384 one, rounds 3-5), so it gets its own short ladder, warmed
    small = {}
    if n_full > 8192:
        X8, _ = syn.make_dataset(8192, d)
        K8 = orc.covariance_matrix("se_ard", th, X8)
        for i, t in enumerate(sorted({t for t in (4, 8, 16, 32) if t <= max(probe)} or {threads})):
            with limited(t):
                for rep in range(2 if i == 0 else 1):
                    Kc = np.asfortranarray(K8)
                    t0 = time.perf_counter()
                    sla.lu_factor(Kc, overwrite_a=True, check_finite=False)
                    small[t] = time.perf_counter() - t0
        del K8, Kc
    n_ref = max(measured) if measured else None
    predicted = None
    if n_ref is not None:
        predicted = measured[n_ref][0] * (n_full / n_ref) ** 2 + measured[n_ref][1] * (n_full / n_ref) ** 3
    if predicted is None or predicted <= max_full_s:
        tb, tf, ll = timed(n_full, threads)
        measured[n_full] = (tb, tf, ll)
        est = tb + tf
        how = (f"MEASURED: one whole evaluation at N={n_full} (build on a {build_threads}-thread pool {tb:.2f} s, LU+solve on "
               f"{threads} LAPACK threads {tf:.2f} s = {est:.1f} s/eval)")
    else:
        est = predicted
        how = (f"N={n_full} NOT measured (the N={n_ref} sample predicts {predicted:.0f} s > {max_full_s:.0f} s): scaled from the "
               f"measured N={n_ref} evaluation by (N/{n_ref})^2 (build) and ^3 (LU)")

    with limited(threads):                       # information only: SPD-aware variant, and cfg 1 as is
        X, y = syn.make_dataset(4096, d)
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
    return {"value": 1.0 / est, "unit": "evals/s", "cores": int(max(threads, build_threads)), "kind": "port",
            "threads": {"lu": int(threads), "build_pool": int(build_threads),
                        "lu_ladder_at_N": int(n_lad), "lu_ladder_s": {str(k): round(v, 3) for k, v in sorted(probe.items())}},
            "sample": "CPU oracle (numpy build + scipy dgetrf/dgetrs LU restatement of BGP:29-43,130-141,181-199; not "
                      f"Mathematica), d={d}, whole evaluations. " + how,
            "measured": {f"N{n}": {"build_s": round(v[0], 3), "lu_solve_s": round(v[1], 3), "loglik": v[2]}
                         for n, v in sorted(measured.items())},
            "also": {"predicted_s_at_n_full_from_smaller_sample": None if predicted is None else round(predicted, 1),
                     "cholesky_variant_s_at_N4096": round(t_chol, 4),
                     "cfg1_N512_d1_evals_per_s": round(1.0 / t_cfg1, 2),
                     "lu_s_at_N8192_by_threads": {str(k): round(v, 3) for k, v in sorted(small.items())},
                     "host_logical_cpus": os.cpu_count()}}


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


def under_profiler() -> bool:
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def _pmc_pass(counters, n: int, d: int, device: int, probe: str = "--pmc-probe"):
    """One child process (`bench.py --pmc-probe`: this file, no torch, two evaluations) under `rocprofv3 --pmc <counters>` --
    counters only, no trace domain next to them, as MI355X_MICROARCH.md prescribes.  Returns the rows of the counter CSV, or
    None when rocprofv3 is missing, this process is itself being profiled, or the pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or under_profiler():
        return None
    tmp = tempfile.mkdtemp(prefix="gphip_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp", GPHIP_NO_TORCH="1", HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(device)),
                   LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", tmp, "-o", "p", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), probe, "--npoints", str(n), "--dim", str(d)]
        res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
        if res.returncode != 0:
            return None
        rows = []
        for path in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                rows.extend(csv.DictReader(f))
        return rows
    except Exception:                                           # never let the counters break the headline
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_pmc_traffic(n: int, d: int, device: int, kernel_substr: str = "gemm_nt_kernel<double, 0,"):
    """HBM-side bytes per launch of the dominant kernel measured IN THIS RUN: two passes (FETCH_SIZE, WRITE_SIZE: the TCC
    block cannot hold both); FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request on wide coalesced reads), both are
    KiB.  Returns None (and the caller falls back to the committed profiles/ summary) when a pass is not possible."""
    got = {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = _pmc_pass([counter], n, d, device)
        if rows is None:
            return None
        vals = [float(r["Counter_Value"]) for r in rows if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
        if not vals:
            return None
        got[counter] = (sum(vals) / len(vals), len(vals))
    return {"bytes_per_launch": (2.0 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024.0,
            "fetch_kib_per_launch": got["FETCH_SIZE"][0], "write_kib_per_launch": got["WRITE_SIZE"][0],
            "launches": got["FETCH_SIZE"][1], "seconds": round(time.perf_counter() - t0, 1)}


def live_kbuild_clock(n: int, d: int, device: int, kernel_substr: str = "kbuild_mfma_kernel<double"):
    """What the kernel build's HBM fraction depends on besides the code: the SHADER CLOCK it runs at.  Up to round 4 the build
    was co-limited by fp64 VALU issue and the HBM store rate (profiles/r03_kbuild_pmc.md), so its time moved with the clock the
    box sustains; since round 5 the distance loop runs on the matrix pipe (kbuild_mfma_kernel) and the VALU share is what this
    probe shows.
    One more PMC pass of the probe: SQ_BUSY_CYCLES (summed over the 32 shader engines) / 32 / launch duration = shader
    clock; SQ_INSTS_VALU x 4 cycles (fp64: 16 lanes per SIMD per clock) / (1024 SIMDs x cycles) = fp64-VALU-busy fraction.
    The probe's first build launch is the process's first large kernel ("cold": clocks still high after idle), the second
    follows ~0.2 s of MFMA work -- the situation of every launch inside the timed loop."""
    rows = _pmc_pass(["SQ_BUSY_CYCLES", "SQ_INSTS_VALU"], n, d, device)
    if rows is None:
        return None
    per = {}
    for r in rows:
        if kernel_substr in r["Kernel_Name"]:
            e = per.setdefault(r["Dispatch_Id"], {"start": float(r["Start_Timestamp"]), "end": float(r["End_Timestamp"])})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
    launches = [v for _, v in sorted(per.items(), key=lambda kv: kv[1]["start"]) if "SQ_BUSY_CYCLES" in v and "SQ_INSTS_VALU" in v]
    if not launches:
        return None
    alg = 8.0 * (n * (n + 1) / 2 + n * d)
    out = []
    for v in launches:
        ns = max(v["end"] - v["start"], 1.0)
        cyc = v["SQ_BUSY_CYCLES"] / 32.0
        out.append({"ms": ns * 1e-6, "shader_clock_ghz": cyc / ns, "fp64_valu_busy": v["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc),
                    "hbm_frac": alg / (ns * 1e-9) / 1e9 / HBM_PEAK_GBS})
    res = {"cold_first_launch": out[0], "after_mfma_work": out[-1], "launches": len(out),
           "note": "one rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU pass over bench.py --pmc-probe (outside the timed region; "
                   "profiled passes run at slightly lower clocks than the timed loop)"}
    # the same pass saw every trailing-SYRK launch of the two evaluations: the shader clock the MFMA-bound kernel sustains (the
    # 78.6 TFLOP/s peak assumes 2.4 GHz; a power-limited chip holds less under back-to-back fp64 MFMAs)
    sy = {}
    for r in rows:
        if "gemm_nt_kernel<double, 0," in r["Kernel_Name"] and r["Counter_Name"] == "SQ_BUSY_CYCLES":
            e = sy.setdefault(r["Dispatch_Id"], {"ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), "cyc": 0.0})
            e["cyc"] += float(r["Counter_Value"]) / 32.0
    big = [e for e in sy.values() if e["ns"] > 1e6]                  # launches of >= 1 ms: the bulk of the factorisation
    if big:
        ns, cyc = sum(e["ns"] for e in big), sum(e["cyc"] for e in big)
        res["syrk"] = {"launches": len(big), "shader_clock_ghz": cyc / ns, "fp64_mfma_peak_at_that_clock_tflops": FP64_MFMA_PEAK_TFLOPS * (cyc / ns) / 2.4}
    return res


def pmc_probe(n: int, d: int) -> None:
    """Child of live_pmc_traffic: the timed loop's evaluation (same data, same theta stream), twice, nothing else."""
    from bayesianinference_amd import _lib, synthetic as syn
    X, y = syn.make_dataset(n, d)
    h = _lib.Handle(X, y, "se_ard")
    base = syn.default_theta("se_ard", d)
    jit = syn.uniform(syn.STREAM_THETA, 1000, 2 * (d + 2))
    for th in base[None, :] * (1.0 + 0.05 * (jit.reshape(2, d + 2) - 0.5)):
        ll, info = h.loglik(th)
        if info != 0:
            raise SystemExit("pmc probe: evaluation failed")
    h.close()


SEPARATOR_KERNEL = "kbuild_kernel<double, 0, 2>"        # what a Matern-3/2 handle's covariance() launches: marks config boundaries


def pmc_probe_mid() -> None:
    """Child of mid_config_clocks: cfg 1, cfg 2 and a cfg-4 batch one after the other, a separator launch between them."""
    from bayesianinference_amd import _lib, synthetic as syn
    Xm, ym = syn.make_dataset(128, 2)
    sep = _lib.Handle(Xm, ym, "matern32_ard")
    ths = syn.default_theta("se_ard", 2)

    def mark():
        sep.covariance(ths)
    for n, d, kernel, reps in ((512, 1, "se", 40), (8192, 8, "se_ard", 12)):
        X, y = syn.make_dataset(n, d)
        h = _lib.Handle(X, y, kernel)
        th = syn.default_theta(kernel, d)
        h.loglik(th); h.loglik(th)
        mark()
        for _ in range(reps):
            h.loglik(th)
        mark()
        h.close()
    X, y = syn.make_dataset(4096, 8)
    Th = syn.theta_batch(200, "se_ard", 8)
    Th[:, -1] = np.maximum(Th[:, -1], 0.05)
    h = _lib.Handle(X, y, "se_ard")
    h.loglik_batch(Th)
    mark()
    h.loglik_batch(Th)
    mark()
    h.close()
    sep.close()


def mid_config_clocks(device: int):
    """Shader clock the chip holds while it runs cfg 1 / cfg 2 / the cfg-4 batch (one rocprofv3 --pmc SQ_BUSY_CYCLES pass over
    bench.py --pmc-probe-mid): sum of SQ_BUSY_CYCLES / 32 shader engines over the config's launches / sum of their durations.
    Lets a reader tell a slower box (lower clock, same code) from slower code (same clock)."""
    rows = _pmc_pass(["SQ_BUSY_CYCLES"], 0, 0, device, probe="--pmc-probe-mid")
    if rows is None:
        return None
    per = {}
    for r in rows:
        if r["Counter_Name"] != "SQ_BUSY_CYCLES":
            continue
        e = per.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), "cyc": 0.0})
        e["cyc"] += float(r["Counter_Value"]) / 32.0
    seq = [per[k] for k in sorted(per)]
    segs, cur, inside = [], [], False
    for e in seq:
        if SEPARATOR_KERNEL in e["name"]:
            if inside:
                segs.append(cur)
            cur, inside = [], not inside
        elif inside:
            cur.append(e)
    names = ["cfg1_n512_d1_f64", "cfg2_n8192_d8_f64", "cfg4_batch_200x4096_f64"]
    if len(segs) != len(names):
        return None
    out = {}
    for nm, seg in zip(names, segs):
        big = [e for e in seg if e["ns"] > 2.0e4] or seg       # launches of >= 20 us carry the clock; tiny ones are dispatch-bound
        ns, cyc = sum(e["ns"] for e in big), sum(e["cyc"] for e in big)
        if ns > 0:
            out[nm] = {"shader_clock_ghz": cyc / ns, "launches": len(seg), "kernel_ms_total": sum(e["ns"] for e in seg) * 1e-6}
    return out


def other_configs(local_rank: int) -> dict:
    """Short measurements of the other BASELINE.json configs on one GPU (reported next to the headline,
    never part of `value`): cfg 1 / cfg 2 = one theta at a time at N=512 d=1 and N=8192 d=8 (the
    reference's sequential-chain usage, latency bound), cfg 4 = 200 theta x N=4096 batched, cfg 5 =
    Matern-5/2 N=65536 d=16 fp32 fit + prediction on 10k test points."""
    from bayesianinference_amd import _lib, synthetic as syn
    out = {}
    try:
        for name, n, d, kernel, reps in (("cfg1_n512_d1_f64", 512, 1, "se", 400), ("cfg2_n8192_d8_f64", 8192, 8, "se_ard", 60)):
            X, y = syn.make_dataset(n, d)
            th = syn.default_theta(kernel, d)
            h = _lib.Handle(X, y, kernel, device=local_rank)
            for _ in range(5):
                h.loglik(th)
            # every evaluation timed on its own (host clock around the blocking call = what a sequential MCMC chain pays,
            # BS:707-745): median and minimum say what the code does, the mean what a noisy box adds
            ts = np.empty(reps)
            for i in range(reps):
                t0 = time.perf_counter()
                _, info = h.loglik(th)
                ts[i] = time.perf_counter() - t0
            # the same evaluation between HIP events on the library's stream (profile class eval_total: no host time)
            h.set_option("profile", 1)
            h.loglik(th)
            h.reset_profile()
            for _ in range(20):
                h.loglik(th)
            ev = h.profile()["eval_total"]
            h.set_option("profile", 0)
            med = float(np.median(ts))
            out[name] = {"ms_per_eval": med * 1e3, "evals_per_s": 1.0 / med, "info": int(info), "reps": int(reps),
                         "ms_min": float(ts.min()) * 1e3, "ms_mean": float(ts.mean()) * 1e3, "ms_p90": float(np.quantile(ts, 0.9)) * 1e3,
                         "ms_hip_events": ev["ms"] / max(ev["launches"], 1),
                         "cholesky_tflops_at_median": n ** 3 / 3.0 / med / 1e12,
                         "timing": "median of per-evaluation host-clock times (blocking call); ms_hip_events = device time of the "
                                   "same evaluation between HIP events on the library's stream"}
            if n == 8192:
                # the rows either side of the likelihood at the same size: fit + prediction of 100 test points (a7) and
                # likelihood + gradient (f3), median of 10 blocking calls each
                def med_ms(f, k=10):
                    f()
                    tt = np.empty(k)
                    for i in range(k):
                        t0 = time.perf_counter()
                        f()
                        tt[i] = time.perf_counter() - t0
                    return float(np.median(tt)) * 1e3
                Xs = syn.make_test_points(100, d)
                out[name]["next_rows"] = {"fit_ms": med_ms(lambda: h.fit(th)), "predict_100_points_ms": med_ms(lambda: h.predict(Xs)),
                                          "loglik_and_gradient_ms": med_ms(lambda: h.loglik_grad(th))}
                # "Inverse"[vector] (BGP:194, 407-412): one right-hand side through the single-vector substitution launches
                # (csrc/gp_trsv.h; the factor streamed once per triangle: 2 x s N^2 / 2 bytes) and through the GEMM-shaped one
                h.fit(th)
                bvec = syn.normal(syn.STREAM_NOISE, 0, n)
                one = med_ms(lambda: h.solve(bvec), 20)
                h.set_option("trsv", 0)
                old = med_ms(lambda: h.solve(bvec), 5)
                h.set_option("trsv", 1)
                out[name]["next_rows"]["solve_one_vector_ms"] = one
                out[name]["next_rows"]["solve_one_vector_gemm_path_ms"] = old
                out[name]["next_rows"]["solve_one_vector_frac_of_hbm_time"] = (8.0 * n * n / HBM_PEAK_GBS / 1e9 * 1e3) / one
            h.close()
    except Exception as exc:                                    # never let an extra break the headline
        out["cfg1_cfg2_error"] = repr(exc)
    try:
        # batches of thetas at the small sizes (the initial sweep of the reference's sampler, BS:902-916, is such a batch): all
        # slots of a call share ONE dataflow launch while they have <= 34 000 tile tasks together (round 6, option
        # dataflow_max_tasks; profiles/r06_batch_crossover.txt)
        for name, n, d, kernel, B in (("cfg1_batch_400x512_d1_f64", 512, 1, "se", 400), ("batch_32x1024_d8_f64", 1024, 8, "se_ard", 32)):
            X, y = syn.make_dataset(n, d)
            Th = syn.theta_batch(B, kernel, d)
            Th[:, -1] = np.maximum(Th[:, -1], 0.05)
            h = _lib.Handle(X, y, kernel, device=local_rank)
            h.loglik_batch(Th); h.loglik_batch(Th)
            ts = np.empty(9)
            for i in range(9):
                t0 = time.perf_counter()
                _, info = h.loglik_batch(Th)
                ts[i] = time.perf_counter() - t0
            med = float(np.median(ts))
            out[name] = {"evals_per_s": B / med, "ms_per_batch_median": med * 1e3, "tflops": B * n ** 3 / 3.0 / med / 1e12,
                         "failed": int((info != 0).sum())}
            h.close()
    except Exception as exc:
        out["small_batch_error"] = repr(exc)
    try:
        # the rows either side of the likelihood at the HEADLINE size (a3 / a7 after gphip_fit at N = 32768, d = 8): the factor is
        # 4.3 GB, so K^-1 b of one vector has 2 x 4.3 GB to stream (gp_trsv.h) and a prediction of 100 test points is one forward
        # dataflow launch over it -- medians of blocking calls
        n, d = 32768, 8
        X, y = syn.make_dataset(n, d)
        th = syn.default_theta("se_ard", d)
        h = _lib.Handle(X, y, "se_ard", device=local_rank)

        def med_ms(f, k):
            f()
            tt = np.empty(k)
            for i in range(k):
                t0 = time.perf_counter()
                f()
                tt[i] = time.perf_counter() - t0
            return float(np.median(tt)) * 1e3
        fit = med_ms(lambda: h.fit(th), 3)
        Xs = syn.make_test_points(100, d)
        bvec = syn.normal(syn.STREAM_NOISE, 0, n)
        one = med_ms(lambda: h.solve(bvec), 8)
        out["cfg3_rows_n32768_d8_f64"] = {"fit_ms": fit, "predict_100_points_ms": med_ms(lambda: h.predict(Xs), 5), "solve_one_vector_ms": one,
                                          "solve_one_vector_tb_per_s_of_factor": 8.0 * n * n / (one * 1e-3) / 1e12,
                                          "what": "gphip_fit / gphip_predict (100 test points) / gphip_solve (one vector) at the headline size"}
        h.close()
    except Exception as exc:
        out["cfg3_rows_error"] = repr(exc)
    try:
        X, y = syn.make_dataset(4096, 8)
        Th = syn.theta_batch(200, "se_ard", 8)
        Th[:, -1] = np.maximum(Th[:, -1], 0.05)
        h = _lib.Handle(X, y, "se_ard", device=local_rank)
        h.loglik_batch(Th[:8])
        h.loglik_batch(Th)
        ts = np.empty(5)
        for i in range(5):
            t0 = time.perf_counter()
            _, info = h.loglik_batch(Th)
            ts[i] = time.perf_counter() - t0
        dt = float(np.median(ts))
        # per launch class: HIP events of the library's profile mode 2 (EVERY launch between its own event pair) with the
        # look-ahead schedule OFF for this one extra batch -- one stream, nothing overlaps, so a class's time is its own and
        # the classes add up to that batch's time (with look-ahead on, launches of the two streams share the chip and the
        # bracketed times double count)
        h.set_option("lookahead", 0)
        h.set_option("profile", 2)
        h.loglik_batch(Th)
        h.reset_profile()
        t0 = time.perf_counter()
        h.loglik_batch(Th)
        dt_serial = time.perf_counter() - t0
        prof = h.profile()
        h.set_option("profile", 0)
        h.set_option("lookahead", 1)
        classes = {}
        for cls, what, peak, unit in (("syrk_trailing", "trailing SYRK / GEMM updates (gemm_nt_kernel<double, 0, ..>)", FP64_MFMA_PEAK_TFLOPS, "TFLOP/s"),
                                      ("gemm_panel", "in-panel GEMM updates, K = 128 .. 384 (gemm_nt_kernel<double, 1, ..>)", FP64_MFMA_PEAK_TFLOPS, "TFLOP/s"),
                                      ("trsm", "panel solves X <- A W^T (gemm_nt_kernel<double, 2, ..>)", FP64_MFMA_PEAK_TFLOPS, "TFLOP/s"),
                                      ("potrf", "128 x 128 diagonal blocks (potrf128_kernel)", FP64_MFMA_PEAK_TFLOPS, "TFLOP/s"),
                                      ("kbuild", "kernel-matrix build", HBM_PEAK_GBS, "GB/s")):
            r = _roof(prof[cls], peak, unit, what)
            if r:
                r["ms_total"] = prof[cls]["ms"]
                classes[cls] = r
        out["cfg4_batch_200x4096_f64"] = {"evals_per_s": 200 / dt, "tflops": 200 * 4096 ** 3 / 3 / dt / 1e12,
                                          "frac_of_fp64_mfma_peak": 200 * 4096 ** 3 / 3 / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                          "failed": int((info != 0).sum()), "ms_per_batch_median": dt * 1e3, "ms_per_batch_min": float(ts.min()) * 1e3,
                                          "reps": 5, "ms_hip_events": prof["eval_total"]["ms"] / max(prof["eval_total"]["launches"], 1),
                                          "classes": classes,
                                          "classes_serial_batch_ms": dt_serial * 1e3,
                                          "classes_note": "measured in ONE extra batch with the look-ahead schedule off (single stream): "
                                                          "exclusive times per launch class; their sum is that batch's time, which the "
                                                          "two-stream schedule of the timed batches beats by overlapping the classes"}
        h.close()
    except Exception as exc:                                    # never let an extra break the headline
        out["cfg4_error"] = repr(exc)
    try:
        # cfg 4 as BASELINE.json words it: the SAMPLER with 200 live points over (l, sigma_f, sigma_n) on an N=4096 problem --
        # the native batched driver (gphip_nested_sampling: 200 lock-step walkers x 20 Metropolis steps per round, every
        # round ONE gphip_loglik_batch of 200 thetas), 150 nested iterations after the 200-point initial sweep
        X, y = syn.make_dataset(4096, 1)
        h = _lib.Handle(X, y, "se", device=local_rank)
        box = [(0.1, 10.0), (0.1, 10.0), (0.01, 1.0)]
        h.loglik_batch(np.tile(np.array([[1.0, 1.0, 0.1]]), (200, 1)))             # size the 200-slot workspace first
        t0 = time.perf_counter()
        res = h.nested_sampling(box, pool=200, walkers=200, mc_steps=20, max_iterations=150, min_iterations=150, seed=1)
        dt = time.perf_counter() - t0
        out["cfg4_nested_sampling_200live_n4096_f64"] = {
            "likelihood_evals_per_s": res["LikelihoodEvaluations"] / dt, "likelihood_evals": int(res["LikelihoodEvaluations"]),
            "nested_iterations": int(res["GeneratedNestedSamples"]), "seconds": dt, "crude_log_evidence": res["CrudeLogEvidence"]}
        h.close()
    except Exception as exc:
        out["cfg4_ns_error"] = repr(exc)
    try:
        n, d, m = 65536, 16, 10000
        X, y = syn.make_dataset(n, d)
        th = syn.default_theta("matern52_ard", d, dtype="f32")
        h = _lib.Handle(X, y, "matern52_ard", dtype=32, device=local_rank)
        h.loglik(th)
        h.set_option("profile", 1)
        h.reset_profile()
        t0 = time.perf_counter()
        info = h.fit(th)
        tf = time.perf_counter() - t0
        t0 = time.perf_counter()
        mu, var = h.predict(syn.make_test_points(m, d))
        tp = time.perf_counter() - t0
        prof = h.profile()
        out["cfg5_matern52_n65536_d16_f32"] = {
            "fit_ms": tf * 1e3, "cholesky_tflops": n ** 3 / 3 / tf / 1e12, "predict_10k_ms": tp * 1e3, "info": int(info),
            "finite": bool(np.all(np.isfinite(mu)) and np.all(var > 0)),
            "roofline_syrk_f32": _roof(prof["syrk_trailing"], FP32_MFMA_PEAK_TFLOPS, "TFLOP/s",
                                       "gemm_nt_kernel<float, 0, 2, 2, 2> (trailing SYRK, v_mfma_f32_16x16x4_f32)"),
            "roofline_kbuild_f32": _roof(prof["kbuild"], HBM_PEAK_GBS, "GB/s", "kbuild_mfma_kernel<float, 4, 1> (Matern-5/2, d=16; cross term of the squared distances on v_mfma_f32_16x16x4_f32)"),
            "roofline_predict_epilogue_f32": _roof(prof["predict_epilogue"], HBM_PEAK_GBS, "GB/s",
                                                   "predict_partial_kernel<float> + predict_finish_kernel (V streamed once)")}
        h.close()
    except Exception as exc:
        out["cfg5_error"] = repr(exc)
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--npoints", dest="n", type=int, default=32768)     # (--npoints: torchrun's own parser chokes on an abbreviable "--n")
    ap.add_argument("--d", "--dim", dest="d", type=int, default=8)              # (--dim: same reason)
    ap.add_argument("--panel", type=int, default=0)
    ap.add_argument("--mode", choices=["theta", "cholesky", "cholesky-torch"], default="theta",
                    help="N>1 only. theta (default): ranks evaluate disjoint theta, no data-path collective, "
                         "weak scaling.  cholesky: ONE evaluation per step sharded over all ranks with the 1-D "
                         "block-cyclic Cholesky (RCCL broadcast of factored panels) run INSIDE the library through "
                         "the C ABI (gphip_create_rank), strong scaling.  cholesky-torch: the same schedule driven by "
                         "the Python harness (dist_cholesky.py, torch.distributed broadcasts).")
    ap.add_argument("--supertile", type=int, default=0, help="experiment: XCD-private 8x8 super-tile order")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alone", action="store_true", help="skip the extra look-ahead-off evaluation behind roofline_syrk_alone "
                                                            "(profiling runs: keeps the rocprof launch statistics those of the timed schedule)")
    ap.add_argument("--no-extras", action="store_true", help="skip the short BASELINE.json cfg-4 / cfg-5 measurements")
    ap.add_argument("--pmc-probe", action="store_true", help=argparse.SUPPRESS)      # child mode of live_pmc_traffic
    ap.add_argument("--pmc-probe-mid", action="store_true", help=argparse.SUPPRESS)  # child mode of mid_config_clocks
    ap.add_argument("--no-live-pmc", action="store_true", help="do not measure roofline.traffic in this run (two rocprofv3 --pmc "
                                                               "passes over a child process); replay profiles/ instead")
    ap.add_argument("--strong-timeout", type=float, default=180.0, help="seconds the strong-scaling series may take before the "
                                                                        "record is printed without it")
    ap.add_argument("--strict-strong", action="store_true", help="N>1: exit with code 3 when the strong-scaling series fails or hangs (tests); by "
                                                                 "default the failure is recorded in the line (`strong.error`) and the run, whose "
                                                                 "weak-scaling headline was measured before the series started, ends with code 0")
    ap.add_argument("--no-strong", action="store_true", help="N>1, mode theta: skip the short strong-scaling series (ONE "
                                                             "factorisation sharded over all ranks) printed as the `strong` sub-record")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    if args.pmc_probe_mid:
        pmc_probe_mid()
        return
    if args.pmc_probe:
        pmc_probe(args.n, args.d)
        return
    import torch
    from bayesianinference_amd import _lib, synthetic as syn

    if not torch.cuda.is_available() or _lib.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (gfx950): the HIP path has no CPU fallback")
    # GPHIP_BENCH_BACKEND=gloo (tests only, tests/test_gpu_bench_multirank.py): the multi-rank control flow of this file on
    # a ONE-GPU box -- every rank on device 0, torch.distributed over gloo, the library's collectives through the
    # tests-only shared-memory library named by $GPHIP_RCCL_PATH (real RCCL refuses two ranks on one device)
    backend = os.environ.get("GPHIP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    red_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    n, d = args.n, args.d
    X, y = syn.make_dataset(n, d)                      # every rank regenerates the same data
    sharded = dist is not None and args.mode in ("cholesky", "cholesky-torch")
    if sharded and args.mode == "cholesky":
        # ONE likelihood factored by all GPUs together, entirely behind the C ABI: rank 0 draws the RCCL id
        # (gphip_comm_unique_id), torch.distributed only hands the 128 bytes round, every rank creates its handle
        # with gphip_create_rank, and gphip_loglik is then a collective call (include/gphip.h)
        box = [_lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        h = _lib.Handle(X, y, "se_ard", device=local_rank, rank=rank, world=world, comm_id=box[0])
        h.set_option("shard_min_n", 0)
    else:
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

    evaluate = h.loglik
    if sharded:
        # every rank steps through the SAME theta
        jit0 = syn.uniform(syn.STREAM_THETA, 1000, total_steps * (d + 2))
        thetas = base[None, :] * (1.0 + 0.05 * (jit0.reshape(total_steps, d + 2) - 0.5))
        if args.mode == "cholesky-torch":              # the Python/torch.distributed harness of the same schedule
            from bayesianinference_amd.dist_cholesky import DistributedCholesky, TorchDistComm
            dc = DistributedCholesky({rank: h}, TorchDistComm(dist), device=local_rank)

            def evaluate(th):
                ll, _, _, info = dc.loglik(th)
                return ll, info

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
        t = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = h.profile()
    bad = [v for v in vals if v[1] != 0 or not np.isfinite(v[0])]
    if bad:
        raise SystemExit(f"bench: evaluation failed: {bad[:3]}")
    # Outside the timed region, rank 0 at N=1: the SAME kernel with the chip to itself.  In the timed region the
    # trailing SYRK shares the CUs with the look-ahead stream (LA update + potrf + panel solves of the next panel), so
    # its event time there includes that contention; one extra evaluation with look-ahead off times it alone.
    alone = None
    if world == 1 and not sharded and not args.no_alone:
        h.set_option("lookahead", 0)
        h.reset_profile()
        evaluate(thetas[0])
        alone = h.profile()["syrk_trailing"]
        h.set_option("lookahead", 1)

    def record(strong, alone):
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
                         "flops": "algorithmic m(m+1) nb per launch (SURVEY.md 8d), summed over the timed launches",
                         "launches": int(syrk["launches"]), "avg_launch_ms": syrk["ms"] / max(syrk["launches"], 1),
                         "traffic": None},
            "cholesky_frac_of_fp64_mfma_peak": chol_flops * args.steps / dt / 1e12 / (world if sharded else 1) / FP64_MFMA_PEAK_TFLOPS,
            # second target of north_star: >= 60 % of the HBM roofline on the kernel-matrix build, measured in THIS run
            "roofline_kbuild": _roof(prof["kbuild"], HBM_PEAK_GBS, "GB/s",
                                     "kbuild_mfma_kernel<double, 2, 0> (SE-ARD d=8, cross term of the squared distances on v_mfma_f64_16x16x4_f64; bytes = 8 [N(N+1)/2 + N d])"),
        }
        out["roofline"]["note"] = ("timed inside the evaluation, where the SYRK shares the chip with the look-ahead stream; "
                                   "roofline_syrk_alone = the same kernel, one extra evaluation with look-ahead off")
        if strong is not None:
            out["strong"] = strong
        if alone is not None:
            out["roofline_syrk_alone"] = _roof(alone, FP64_MFMA_PEAK_TFLOPS, "TFLOP/s",
                                               "gemm_nt_kernel<double, 0, 2, 2, 2>, look-ahead off (outside the timed region)")
        tr = pmc_traffic()
        if tr is not None:
            # the committed PMC passes (profiles/); replaced below by counters measured in THIS run when that is possible
            out["roofline"]["traffic"] = tr["bytes_per_launch"]
            out["roofline"]["traffic_replayed_from"] = "profiles/" + tr["source"]
            out["roofline"]["traffic_unit"] = "HBM-side bytes per launch (rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE, KiB -> B)"
            out["roofline"]["algorithmic_bytes_per_launch"] = syrk["bytes"] / max(syrk["launches"], 1)
        return out

    # N > 1, default (weak, theta-sharded) mode: ALSO a short strong-scaling series -- the split north_star names: ONE
    # likelihood factored by all ranks together (1-D block-cyclic Cholesky, RCCL panel broadcast, entirely behind the C ABI
    # through gphip_create_rank).  Outside the timed region of the headline; reported as the `strong` sub-record.
    strong = None
    strong_failed = False
    if dist is not None and not sharded and not args.no_strong:
        # Watchdog: this is the only place where the bench waits inside RCCL collectives of the library (real multi-rank
        # RCCL has only ever been exercised here).  If one does not return, the headline measured above must not be lost:
        # after --strong-timeout seconds rank 0 prints the record with the failure noted (`strong.error`) and every rank leaves --
        # with exit code 0 by default (the weak-scaling `value` of this run is valid: it was measured before the series started),
        # with code 3 under --strict-strong (tests: a hung collective must not pass silently).  Exactly one record is printed:
        # the main thread and the watchdog both take `emit_lock` and check `emitted` before they print.
        import threading
        emit_lock = threading.Lock()
        emitted = [False]
        partial = [None]                                       # the default series' record once IT is complete (variants follow)
        early = [None]                                         # the OLDEST schedule's timing, taken before the default's (see below)

        def give_up():
            with emit_lock:
                if emitted[0]:
                    return                                     # the series finished while the timer fired
                emitted[0] = True
                if rank == 0:
                    rec = record(None, None)
                    if partial[0] is not None:                 # the default schedule finished: only an optional variant hung
                        st = dict(partial[0])
                        st["variants_error"] = f"a schedule variant did not return within {args.strong_timeout:.0f} s"
                        rec["strong"] = st
                        rec["strong_speedup"], rec["strong_ms_per_eval"] = st["speedup_vs_one_gpu_weak_step"], st["ms_per_eval"]
                        rec["rccl_ranks"] = st["rccl_ranks"]
                    else:
                        rec["strong"] = {"error": f"no result after {args.strong_timeout:.0f} s (a collective call did not return); "
                                                  "the headline above was measured before this series started"}
                        rec["strong_speedup"] = rec["strong_ms_per_eval"] = rec["rccl_ranks"] = None
                        if early[0] is not None:               # .. but the oldest schedule had returned: the default is what hung
                            rec["strong"]["oldest_schedule"] = early[0]
                            rec["strong"]["error"] = (f"the library's default schedule did not return within {args.strong_timeout:.0f} s; "
                                                      "the round-5 schedule (oldest_schedule) had completed before it")
                    print(json.dumps(rec), flush=True)
                # (the failure must not be visible in the JSON line only)
                print(f"bench: rank {rank}: the optional strong-scaling series did not return within {args.strong_timeout:.0f} s; "
                      "leaving the process (the weak-scaling headline above was measured before it started)", file=sys.stderr, flush=True)
                os._exit(3 if args.strict_strong else 0)
        dog = threading.Timer(args.strong_timeout + (0.0 if rank == 0 else 5.0), give_up)
        dog.daemon = True
        dog.start()
        try:
            box = [_lib.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            hs = _lib.Handle(X, y, "se_ard", device=local_rank, rank=rank, world=world, comm_id=box[0])
            hs.set_option("shard_min_n", 0)
            jit0 = syn.uniform(syn.STREAM_THETA, 1000, 8 * (d + 2))
            ths = base[None, :] * (1.0 + 0.05 * (jit0.reshape(8, d + 2) - 0.5))
            # first the schedule with the longest record in the tests (round 5: per-tile-column panel launches, plain ncclBroadcast,
            # no stream-ordered waits, nothing yields): if the default below should not return on hardware nobody has run it on,
            # the watchdog still has one strong-scaling time to print
            old_opts = {"dist_panel_df": 0, "bcast_two_hop": 0, "dist_owner_yield": 0}
            for k_, v_ in old_opts.items():
                hs.set_option(k_, v_)
            hs.loglik(ths[0]); hs.loglik(ths[1])
            barrier()
            t0s = time.perf_counter()
            sv0 = [hs.loglik(ths[2 + i]) for i in range(5)]
            barrier()
            ts0 = torch.tensor([time.perf_counter() - t0s], device=red_dev, dtype=torch.float64)
            dist.all_reduce(ts0, op=dist.ReduceOp.MAX)
            early[0] = {"options": old_opts, "ms_per_eval": float(ts0.item()) / 5 * 1e3,
                        "speedup_vs_one_gpu_weak_step": (dt / args.steps) / (float(ts0.item()) / 5),
                        "all_ok": bool(all(v[1] == 0 and np.isfinite(v[0]) for v in sv0))}
            for k_ in old_opts:
                hs.set_option(k_, -1)                          # back to the library's own choice
            hs.loglik(ths[0]); hs.loglik(ths[1])
            barrier()
            t1 = time.perf_counter()
            sv = [hs.loglik(ths[2 + i]) for i in range(5)]
            barrier()
            ds = time.perf_counter() - t1
            ts = torch.tensor([ds], device=red_dev, dtype=torch.float64)
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            ci = hs.comm_info()
            strong = {"what": f"ONE evaluation per step factored by all {world} ranks together (1-D block-cyclic Cholesky, panels "
                              "factored as one dataflow launch by their owner and exchanged over RCCL, gphip_create_rank)",
                      "scaling": "strong", "steps": 5, "ms_per_eval": float(ts.item()) / 5 * 1e3,
                      "evals_per_s": 5 / float(ts.item()), "speedup_vs_one_gpu_weak_step": (dt / args.steps) / (float(ts.item()) / 5),
                      "cholesky_tflops_total": n ** 3 / 3.0 * 5 / float(ts.item()) / 1e12,
                      "rccl_ranks": ci["world"], "comm": ci["comm"], "factor_bytes_per_rank": hs.factor_bytes(),
                      "all_ok": bool(all(v[1] == 0 and np.isfinite(v[0]) for v in sv))}
            strong["oldest_schedule"] = early[0]
            try:
                # what scripts/scale_model.py predicts for THIS world size and the default schedule from single-GPU step times
                # (profiles/r06_owner_path_32768.json) -- printed so that the first multi-GPU run falsifies or
                # confirms it in the same line
                import importlib.util
                here = os.path.dirname(os.path.abspath(__file__))
                spec = importlib.util.spec_from_file_location("scale_model", os.path.join(here, "scripts", "scale_model.py"))
                sm = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(sm)
                with open(os.path.join(here, "profiles", "r06_owner_path_32768.json")) as f:
                    steps_data = json.load(f)
                if steps_data["N"] == n and world in (2, 4, 8):
                    kw = dict(chunks=True, first_ready=0.36, two_hop=world >= 4, owner_yield=world >= 4, issue_us=24.0)
                    strong["model_ms_per_eval"] = {f"alpha_{a:.0f}us_beta_{b:.0f}GBs": sm.simulate(steps_data, "df2_fuse0", world, a, b, **kw)[0] / 1e3
                                                   for a, b in ((10.0, 120.0), (20.0, 60.0), (40.0, 30.0))}
                    strong["model_note"] = ("scripts/scale_model.py on single-GPU step times of an MI355X build box; alpha = latency per "
                                            "collective, beta = bandwidth per receiver; made before any multi-GPU run existed")
            except Exception as exc:
                strong["model_error"] = repr(exc)
            strong["oldest_schedule"]["agrees_with_default"] = bool(all(a[1] == b[1] and abs(a[0] - b[0]) <= 1e-10 * abs(a[0]) for a, b in zip(sv, sv0)))
            strong_failed = not strong["all_ok"]
            # The same five evaluations under every explicit schedule, which only a real multi-GPU node can rank (none is measurable
            # on the one-GPU boxes this code was developed on).  The default above is the library's own choice (round 6, by the
            # model of scripts/scale_model.py): the owner factors its panel AND applies the look-ahead update in ONE dataflow
            # launch that counts finished tiles per tile column, and the broadcast stream waits on those counters
            # (hipStreamWaitValue32), so every column travels while the launch still runs ("dist_panel_df" = 3); from 4 ranks on
            # every panel message goes out as scatter + in-place all-gather ("bcast_two_hop": all links of the xGMI mesh carry
            # 1 / world of a message at once instead of one ring).  The variants: 0 = panels factored tile column by tile column
            # with separate launches, each column broadcast as it becomes final (the round-5 schedule); 2 = the dataflow launch
            # without the counters (its columns are final when it ends).  Results must agree: bit for bit between broadcast
            # forms and between 2 and 3, to 1e-10 relative between 0 and the dataflow launches (summation order inside 64-blocks).
            default_df = int(hs.get_option("last_dist_panel_df"))          # what the library resolved -1 to on this device
            default_hop = int(world >= 4)
            variants = [("per_column_broadcast", {"dist_panel_df": 0, "bcast_two_hop": 0}), ("dist_panel_df", {"dist_panel_df": 2, "bcast_two_hop": 0}),
                        ("column_signals", {"dist_panel_df": 3, "bcast_two_hop": 0})]
            if world > 2:
                variants += [("two_hop", {"dist_panel_df": 0, "bcast_two_hop": 1}), ("two_hop_dist_panel_df", {"dist_panel_df": 2, "bcast_two_hop": 1}),
                             ("two_hop_column_signals", {"dist_panel_df": 3, "bcast_two_hop": 1})]
            variants = [v for v in variants if v[1] != {"dist_panel_df": default_df, "bcast_two_hop": default_hop}]    # (= the default itself)
            # .. and the default with "dist_owner_yield" flipped: on (the library's choice from 4 ranks) the owner's trailing updates
            # are queued behind its panel launch's end event instead of sharing the GPU with it
            default_yield = int(world >= 4)
            variants.append(("owner_yield_off" if default_yield else "owner_yield_on",
                             {"dist_panel_df": default_df, "bcast_two_hop": default_hop, "dist_owner_yield": 1 - default_yield}))
            strong["default_options"] = {"dist_panel_df": default_df, "bcast_two_hop": default_hop, "dist_owner_yield": default_yield}
            strong["variants"] = {}
            partial[0] = dict(strong)
            best = ("default", strong["ms_per_eval"])
            try:
                for vname, opts in variants:
                    hs.set_option("dist_owner_yield", -1)
                    for k_, v_ in opts.items():
                        hs.set_option(k_, v_)
                    hs.loglik(ths[0])
                    barrier()
                    t2 = time.perf_counter()
                    sv2 = [hs.loglik(ths[2 + i]) for i in range(5)]
                    barrier()
                    ts2 = torch.tensor([time.perf_counter() - t2], device=red_dev, dtype=torch.float64)
                    dist.all_reduce(ts2, op=dist.ReduceOp.MAX)
                    exact = (opts["dist_panel_df"] >= 2) == (default_df >= 2)   # (same arithmetic as the default: only hand-over / broadcast form differ)
                    same = bool(all(a[1] == b[1] and (a[0] == b[0] if exact else abs(a[0] - b[0]) <= 1e-10 * abs(a[0])) for a, b in zip(sv, sv2)))
                    ms2 = float(ts2.item()) / 5 * 1e3
                    strong["variants"][vname] = {"options": opts, "ms_per_eval": ms2, "speedup_vs_one_gpu_weak_step": (dt / args.steps) / (ms2 / 1e3),
                                                 "same_results": same}
                    # (a variant that disagrees is recorded, not fatal: the default schedule above is what the run is judged on)
                    if same and ms2 < best[1]:
                        best = (vname, ms2)
            except Exception as exc:                            # (the library fails a collective call on ALL ranks together)
                strong["variants_error"] = repr(exc)
            strong["best_variant"] = best[0]
            if "two_hop" in strong["variants"]:                    # (round-3 field names, kept for readers of older lines)
                strong["two_hop_ms_per_eval"] = strong["variants"]["two_hop"]["ms_per_eval"]
                strong["two_hop_speedup_vs_one_gpu_weak_step"] = strong["variants"]["two_hop"]["speedup_vs_one_gpu_weak_step"]
                strong["two_hop_identical_results"] = strong["variants"]["two_hop"]["same_results"]
            hs.close()
        except Exception as exc:                                    # never let the extra break the headline
            strong = {"error": repr(exc)}
            strong_failed = True
        dog.cancel()
        with emit_lock:
            if emitted[0]:                                          # the watchdog got there first and is ending the process
                time.sleep(60)
            emitted[0] = True

    if rank == 0:
        out = record(strong, alone)
        if world > 1 and not sharded:
            # the split north_star names, next to the weak (theta-sharded) `value`: top-level so that a reader of the line
            # does not mistake the by-construction weak number for it
            ok = strong is not None and "error" not in strong
            out["strong_speedup"] = strong["speedup_vs_one_gpu_weak_step"] if ok else None
            out["strong_ms_per_eval"] = strong["ms_per_eval"] if ok else None
            out["rccl_ranks"] = strong["rccl_ranks"] if ok else None
            # the fastest schedule variant whose results agreed with the default's (the default itself if none beat it)
            bv = strong.get("best_variant", "default") if ok else None
            out["strong_best_variant"] = bv
            out["strong_best_speedup"] = (strong["variants"][bv]["speedup_vs_one_gpu_weak_step"] if ok and bv in strong.get("variants", {})
                                          else out["strong_speedup"])
        if world == 1 and not args.no_extras and not args.no_live_pmc and args.mode == "theta":
            live = live_pmc_traffic(n, d, local_rank)
            if live is not None:
                committed = out["roofline"].get("traffic")
                out["roofline"]["traffic"] = live["bytes_per_launch"]
                out["roofline"].pop("traffic_replayed_from", None)
                out["roofline"]["traffic_unit"] = "HBM-side bytes per launch (rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE, KiB -> B)"
                out["roofline"]["algorithmic_bytes_per_launch"] = prof["syrk_trailing"]["bytes"] / max(prof["syrk_trailing"]["launches"], 1)
                out["roofline"]["traffic_source"] = (
                    f"measured in this run: two child processes of this bench (bench.py --pmc-probe, 2 evaluations each) under "
                    f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, {live['launches']} launches averaged, {live['seconds']} s "
                    f"outside the timed region; committed profiles/ summary says {committed}")
        if world == 1 and not args.no_extras and out.get("roofline_kbuild"):
            # The build is a pure streaming WRITE of 4.3 GB: what the same box does with the plainest possible store stream of
            # the same size (the runtime's own fill kernel behind Tensor.zero_()), measured here with HIP events, is the
            # ceiling it runs against on THIS box -- the 8 TB/s of `peak` is the read-side headline figure.
            try:
                nbytes = int(out["roofline_kbuild"]["algorithmic_per_launch"])
                buf = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{local_rank}")
                buf.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                best = 1e30
                for _ in range(4):
                    e0.record(); buf.zero_(); e1.record(); e1.synchronize()
                    best = min(best, e0.elapsed_time(e1))
                del buf
                gbs = nbytes / (best * 1e-3) / 1e9
                out["roofline_kbuild"]["store_ceiling"] = {
                    "what": "fill kernel (Tensor.zero_) over the same number of bytes on this box, best of 4, HIP events",
                    "achieved": gbs, "unit": "GB/s", "frac_of_peak": gbs / HBM_PEAK_GBS,
                    "kbuild_frac_of_store_ceiling": out["roofline_kbuild"]["achieved"] / gbs}
            except Exception as exc:                                # never let an extra break the headline
                out["roofline_kbuild"]["store_ceiling"] = {"error": repr(exc)}
        if world == 1 and not args.no_extras and not args.no_live_pmc and args.mode == "theta" and out.get("roofline_kbuild"):
            clk = live_kbuild_clock(n, d, local_rank)
            if clk is not None:
                syc = clk.pop("syrk", None)
                out["roofline_kbuild"]["clock_probe"] = clk
                # the same kernel in the probe's PMC pass, where launches run one at a time with idle gaps between them (the
                # clocks recover): `frac` above is the build right behind 180 ms of fp64 MFMA work, at whatever clock the
                # power-limited chip still holds then
                out["roofline_kbuild"]["frac_in_pmc_probe"] = clk["after_mfma_work"]["hbm_frac"]
                if syc is not None:
                    # (the profiled pass serialises kernels: this is the SYRK alone, to be read next to roofline_syrk_alone)
                    syc["note"] = ("shader clock over the trailing-SYRK launches of the PMC probe (SQ_BUSY_CYCLES / 32 shader engines / "
                                   "duration); frac_at_clock = TFLOP/s / (78.6 x clock / 2.4 GHz)")
                    pk = syc["fp64_mfma_peak_at_that_clock_tflops"]
                    syc["roofline_frac_at_clock"] = out["roofline"]["achieved"] / pk
                    if out.get("roofline_syrk_alone"):
                        syc["syrk_alone_frac_at_clock"] = out["roofline_syrk_alone"]["achieved"] / pk
                    out["roofline"]["clock_probe"] = syc
        if world == 1 and not args.no_extras:
            h.close()                                           # free the 8.7 GB workspace first
            out["other_configs"] = other_configs(local_rank)
            if not args.no_live_pmc:
                clocks = mid_config_clocks(local_rank)
                for nm, v in (clocks or {}).items():
                    if nm in out["other_configs"]:
                        out["other_configs"][nm]["clock_probe"] = v
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, d)
        print(json.dumps(out), flush=True)
    h.close()
    if dist is not None:
        dist.destroy_process_group()
    if strong_failed and args.strict_strong:
        sys.exit(3)                                                 # the record is out; the strong series failed: not a success (tests)


if __name__ == "__main__":
    main()

"""Split dataflow launch (option df_split = reserved CUs per XCD for the chain tasks, df_split_width = tiles per column, from the
diagonal down, that count as chain tasks) vs the single launch: ms per single-theta evaluation, results must be bit-identical.
Usage: gpu_df_split.py [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn

sizes = [int(a) for a in sys.argv[1:]] or [4096, 6144, 8192, 10240, 12288]
variants = [("single", {"df_split": 0})]
for cw in (2, 3, 4):
    for res, kib in ((2, 84), (3, 84), (4, 84), (3, 0), (4, 0), (6, 0)):
        variants.append((f"w{cw}r{res}{'x2' if kib == 0 else ''}", {"df_split": res, "df_split_lds_kib": kib, "df_split_width": cw}))
for n in sizes:
    d, kernel = 8, "se_ard"
    X, y = syn.make_dataset(n, d)
    h = _lib.Handle(X, y, kernel)
    th = syn.default_theta(kernel, d)
    ref = None
    for rnd in range(2):
        row = [f"N={n:5d}"]
        for name, opts in variants:
            for k, v in opts.items():
                h.set_option(k, v)
            try:
                h.loglik(th); h.loglik(th)
                reps = 20 if n <= 8192 else 8
                t0 = time.perf_counter()
                for _ in range(reps):
                    ll, info = h.loglik(th)
                dt = (time.perf_counter() - t0) / reps
            except Exception as exc:                                  # noqa: BLE001
                row.append(f"{name}: ERROR {exc}")
                continue
            if ref is None:
                ref = ll
            flag = "" if ll == ref else f" DIFF {ll - ref:.3e}"
            row.append(f"{name}: {dt*1e3:6.3f}{flag}")
        print(" | ".join(row), flush=True)
    h.close()

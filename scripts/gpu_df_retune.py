"""Re-tune of the dataflow schedule's rules after the round-4 fence changes: occupancy (one / two / three workgroups per CU),
parking, single launch vs look-ahead + tail, tail length.  One row per N."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn

def t(h, th, reps):
    h.loglik(th); h.loglik(th)
    t0 = time.perf_counter()
    for _ in range(reps):
        ll, info = h.loglik(th)
    return (time.perf_counter() - t0) / reps * 1e3

mode = sys.argv[1] if len(sys.argv) > 1 else "occ"
if mode == "occ":
    for n in (1024, 2048, 3072, 4096, 5120, 6144, 7168, 8192, 9216, 10240):
        X, y = syn.make_dataset(n, 8); th = syn.default_theta("se_ard", 8)
        h = _lib.Handle(X, y, "se_ard")
        row = [f"N={n:5d}"]
        for name, o in (("auto", {"dataflow_lds_kib": -1, "dataflow_occ3": -1}), ("1/CU", {"dataflow_lds_kib": 84, "dataflow_occ3": 0}),
                        ("2/CU", {"dataflow_lds_kib": 0, "dataflow_occ3": 0}),
                        ("3/CU", {"dataflow_lds_kib": 0, "dataflow_occ3": 1})):
            for k, v in o.items():
                h.set_option(k, v)
            row.append(f"{name}: {t(h, th, 20 if n <= 6144 else 10):6.3f}")
        print(" | ".join(row), flush=True)
        h.close()
else:
    for n in (10240, 12288, 14336, 16384, 20480, 24576, 32768):
        X, y = syn.make_dataset(n, 8); th = syn.default_theta("se_ard", 8)
        h = _lib.Handle(X, y, "se_ard")
        row = [f"N={n:5d}"]
        if n <= 16384:
            h.set_option("dataflow_max_nt", 256); h.set_option("dataflow_fine_nt", 256)
            row.append(f"single launch: {t(h, th, 6):7.3f}")
        h.set_option("dataflow_max_nt", 64); h.set_option("dataflow_fine_nt", 96)
        for tail in (48, 64, 80, 96):
            h.set_option("dataflow_tail", tail)
            h.set_option("dataflow_max_nt", max(64, tail))
            row.append(f"la+tail{tail}: {t(h, th, 6 if n <= 20480 else 3):7.3f}")
        print(" | ".join(row), flush=True)
        h.close()

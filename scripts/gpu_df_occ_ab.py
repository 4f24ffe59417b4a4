import sys, time
sys.path.insert(0, '/root/repo')
from bayesianinference_amd import _lib, synthetic as syn
def t(h, th, reps):
    h.loglik(th); h.loglik(th)
    t0 = time.perf_counter()
    for _ in range(reps): h.loglik(th)
    return (time.perf_counter() - t0) / reps * 1e3
for n in (3584, 4096, 8192, 10240, 11264, 12288):
    X, y = syn.make_dataset(n, 8); th = syn.default_theta("se_ard", 8)
    h = _lib.Handle(X, y, "se_ard")
    h.set_option("panel_df", 0)
    row = [f"N={n}"]
    for rnd in range(2):
        for name, o in (("1/CU", {"dataflow_lds_kib": 84, "dataflow_occ3": 0}), ("2/CU", {"dataflow_lds_kib": 0, "dataflow_occ3": 0}), ("3/CU", {"dataflow_lds_kib": 0, "dataflow_occ3": 1})):
            if name == "1/CU" and n > 5000: continue
            for k, v in o.items(): h.set_option(k, v)
            row.append(f"{name}: {t(h, th, 20 if n <= 6144 else 8):.3f}")
    print(" | ".join(row), flush=True)
    h.close()

import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from bayesianinference_amd import _lib, synthetic as syn
for n, B in ((4096, 200), (2048, 200), (1024, 200), (4096, 32)):
    X, y = syn.make_dataset(n, 8)
    Th = syn.theta_batch(B, "se_ard", 8); Th[:, -1] = np.maximum(Th[:, -1], 0.05)
    h = _lib.Handle(X, y, "se_ard")
    ref, iref = h.loglik_batch(Th)
    def t():
        h.loglik_batch(Th)
        t0 = time.perf_counter()
        for _ in range(3): out, info = h.loglik_batch(Th)
        return (time.perf_counter() - t0) / 3 * 1e3, out, info
    base, _, _ = t()
    row = [f"N={n} B={B}: multi-kernel {base:.2f} ms = {B*n**3/3/base/1e9:.1f} TF"]
    h.set_option("dataflow_max_slots", 256)
    for occ in (0, 1):
        h.set_option("dataflow_occ3", occ); h.set_option("dataflow_lds_kib", 0)
        dt, out, info = t()
        ok = np.array_equal(info, iref) and np.allclose(out, ref, rtol=1e-9)
        row.append(f"dataflow occ3={occ}: {dt:.2f} ms = {B*n**3/3/dt/1e9:.1f} TF ok={ok}")
    print(" | ".join(row), flush=True)
    h.close()

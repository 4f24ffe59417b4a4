"""Phase-shifted batch groups (option batch_groups) for large theta batches: time and BIT-identity against the un-grouped call.
   python scripts/gpu_batch_groups.py [N B] ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
cases = [(4096, 200), (2048, 400), (8192, 48), (4096, 64), (1024, 800)]
if len(sys.argv) > 2:
    cases = [(int(sys.argv[1]), int(sys.argv[2]))]
for n, B in cases:
    X, y = syn.make_dataset(n, 8)
    Th = syn.theta_batch(B, "se_ard", 8)
    Th[:, -1] = np.maximum(Th[:, -1], 0.05)
    Th[3, 0] = np.nan                                   # one bad theta and one not-SPD candidate travel through the groups too
    h = _lib.Handle(X, y, "se_ard")
    ref = None
    for rnd in range(2):
        row = [f"N={n} B={B}"]
        for G in (1, 2, 3, 4, 6):
            h.set_option("batch_groups", G)
            h.set_option("batch_group_min", 8)
            print(f'  N={n} B={B} G={G} ...', flush=True)
            h.loglik_batch(Th)
            h.sync()
            t0 = time.perf_counter(); out, info = h.loglik_batch(Th); dt = time.perf_counter() - t0
            if ref is None:
                ref = (out.copy(), info.copy())
            ok = np.array_equal(info, ref[1]) and np.array_equal(out[info == 0], ref[0][ref[1] == 0])
            row.append(f"G={G}: {dt*1e3:7.2f} ms {B*n**3/3/dt/1e12:5.1f} TF{'' if ok else ' MISMATCH'}")
        print(" | ".join(row), flush=True)
    h.close()

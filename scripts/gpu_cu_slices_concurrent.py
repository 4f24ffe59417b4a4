"""Control for scripts/gpu_cu_partition.py: W INDEPENDENT plain handles, one per CU slice (GPHIP_CU_SLICE="r/W"), evaluating the same
problem at the same time from W host threads.  No sharding, no schedule: if some slices come out slower than others here, the
asymmetry seen in the partitioned sharded runs belongs to the emulation (shared L2 / fabric / dispatcher), not to the schedule.
   python3 scripts/gpu_cu_slices_concurrent.py [N]"""
import os, sys, threading, time
os.environ["GPHIP_TEST_HOOKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
X, y = syn.make_dataset(n, 8)
th = syn.default_theta("se_ard", 8)
for W in (2, 4, 8):
    hs = []
    for r in range(W):
        os.environ["GPHIP_CU_SLICE"] = f"{r}/{W}"
        hs.append(_lib.Handle(X, y, "se_ard"))
    del os.environ["GPHIP_CU_SLICE"]
    alone = []
    for h in hs:
        h.loglik(th)
        t0 = time.perf_counter(); h.loglik(th); alone.append((time.perf_counter() - t0) * 1e3)
    out = [0.0] * W
    go = threading.Barrier(W)
    def work(r):
        go.wait()
        t0 = time.perf_counter()
        for _ in range(3):
            hs[r].loglik(th)
        out[r] = (time.perf_counter() - t0) / 3 * 1e3
    ts = [threading.Thread(target=work, args=(r,)) for r in range(W)]
    [t.start() for t in ts]; [t.join() for t in ts]
    print(f"N={n} W={W}: alone per slice {' '.join(f'{a:.1f}' for a in alone)} ms | all {W} at once {' '.join(f'{a:.1f}' for a in out)} ms", flush=True)
    for h in hs:
        h.close()

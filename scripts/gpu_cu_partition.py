"""Strong scaling of the sharded schedule with ranks that really run SIDE BY SIDE -- on one GPU.  A stream created with a CU
mask only uses the CUs it names.  With GPHIP_CU_PARTITION=1 (developer hook in gphip_multi.inc group_create) the W virtual
ranks of a one-device group get 1 / W of the CUs of every XCD each (256 / W CUs), so their launches overlap in time as they
would on W GPUs -- unlike plain virtual ranks, which time-slice the whole chip.  "One GPU" of this experiment is a plain handle
restricted to one such slice (GPHIP_CU_SLICE="0/W").  What it measures: the schedule's critical path and its overlap
machinery (owner chain, column signals, look-ahead, host issue) with ideal links -- panels are read in place, or copied through
HBM by blit kernels on the masked streams; HBM, the L2s and the Infinity Cache are shared, so the slices are not independent
GPUs (a slice sees MORE bandwidth per CU than a whole chip does).
What it cannot measure: xGMI.   python3 scripts/gpu_xcd_partition.py [N]"""
import os, sys, time
os.environ["GPHIP_TEST_HOOKS"] = "1"          # (the CU-partition hooks are read only with this)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y = syn.make_dataset(n, 8)
th = syn.default_theta("se_ard", 8)


def timeit(h, reps=3):
    h.loglik(th)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = h.loglik(th)
    return (time.perf_counter() - t0) / reps * 1e3, r[0]


h = _lib.Handle(X, y, "se_ard")
full, ll0 = timeit(h)
h.close()
print(f"N={n}: plain handle on the whole chip (256 CUs) {full:8.2f} ms   ll={ll0:.12g}", flush=True)
for W in (2, 4, 8):
    os.environ["GPHIP_CU_SLICE"] = f"0/{W}"
    h = _lib.Handle(X, y, "se_ard")
    one, ll1 = timeit(h, 2)
    h.close()
    del os.environ["GPHIP_CU_SLICE"]
    assert ll1 == ll0
    print(f"W={W}: ONE rank's share of the chip ({256 // W} CUs), plain handle: {one:8.2f} ms = {one / full:.2f} x the whole chip's time", flush=True)
    os.environ["GPHIP_CU_PARTITION"] = "1"
    g = _lib.Handle(X, y, "se_ard", device=[0] * W)
    del os.environ["GPHIP_CU_PARTITION"]
    g.set_option("shard_min_n", 0)
    for label, opts in (("column signals, owner yields (the default from 4 ranks), in place", {"dist_owner_yield": 1}),
                        ("owner does not yield (dist_owner_yield=0), in place", {"dist_owner_yield": 0}),
                        ("the same, panels copied into receive buffers", {"share_local_panels": 0, "dist_owner_yield": 1}),
                        ("dist_panel_df=2 (no signals), owner yields, copies", {"share_local_panels": 0, "dist_panel_df": 2, "dist_owner_yield": 1}),
                        ("dist_panel_df=0 (per-tile-column launches), owner yields, copies", {"share_local_panels": 0, "dist_panel_df": 0, "dist_owner_yield": 1})):
        g.set_option("share_local_panels", 1); g.set_option("dist_panel_df", -1); g.set_option("dist_owner_yield", -1)
        for k_, v_ in opts.items():
            g.set_option(k_, v_)
        t, ll = timeit(g)
        ok = abs(ll - ll0) <= 1e-10 * abs(ll0)
        print(f"   {W} ranks on {256 // W} CUs each, {label:64s} {t:8.2f} ms = {one / t:.2f}x of {W} (vs the one-rank share); "
              f"{full / t:.2f} of the whole chip's plain rate; issue {g.get_option('last_issue_us') / 1e3:.1f} ms; results {'ok' if ok else 'DIFFER'}", flush=True)
    g.close()

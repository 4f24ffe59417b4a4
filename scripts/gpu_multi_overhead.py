"""What the in-library sharded schedule costs on ONE GPU (no xGMI involved): N=32768 evaluated by a plain handle, by a
rank handle over RCCL at world size 1, and by 2 / 4 / 8 virtual ranks sharing the device (same total work, device copies
for the panel exchange).  Differences = packing, unpacking, the broadcast calls, host-side scheduling, lost look-ahead."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y = syn.make_dataset(n, 8)
th = syn.default_theta("se_ard", 8)

def timeit(h, reps=3, fit=False):
    f = (lambda: h.fit(th)) if fit else (lambda: h.loglik(th))
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t0) / reps, r

h = _lib.Handle(X, y, "se_ard")
dt, r = timeit(h); print(f"N={n} plain handle:                {dt*1e3:8.2f} ms  ll={r[0]:.10g}", flush=True)
h.set_option("dataflow_tail", 0)
dt, r = timeit(h); print(f"N={n} plain, no dataflow tail:     {dt*1e3:8.2f} ms  ll={r[0]:.10g}", flush=True)
h.close()
g = _lib.Handle(X, y, "se_ard", device=0, rank=0, world=1, comm_id=_lib.comm_unique_id())
g.set_option("shard_min_n", 0)
dt, r = timeit(g); print(f"N={n} rank handle, RCCL world 1:   {dt*1e3:8.2f} ms  ll={r[0]:.10g}  {g.comm_info()['comm']}", flush=True)
dt, r = timeit(g, fit=True); print(f"N={n}   ... fit (keeps factor):    {dt*1e3:8.2f} ms", flush=True)
g.close()
for w in (2, 4, 8):
    g = _lib.Handle(X, y, "se_ard", device=[0] * w)
    g.set_option("shard_min_n", 0)
    dt, r = timeit(g); print(f"N={n} {w} virtual ranks, panels read in place: {dt*1e3:8.2f} ms  ll={r[0]:.10g}   "
                             f"[host thread issued the whole schedule of all {w} ranks in {g.get_option('last_issue_us') / 1e3:.2f} ms = "
                             f"{g.get_option('last_issue_us') / w / 1e3:.2f} ms per rank]", flush=True)
    for opts in ({"dist_panel_df": 0}, {"dist_panel_df": 2}, {"dist_panel_df": 3}, {"dist_panel_df": 3, "dist_owner_yield": 1 - int(w >= 4)}):
        for k_, v_ in opts.items():
            g.set_option(k_, v_)
        dt2, r2 = timeit(g)
        print(f"N={n}   ... {opts}: {dt2*1e3:8.2f} ms, issue {g.get_option('last_issue_us') / 1e3:.2f} ms", flush=True)
    g.set_option("dist_panel_df", -1); g.set_option("dist_owner_yield", -1)
    dt, r = timeit(g, fit=True); print(f"N={n}   ... fit (factor stays distributed):  {dt*1e3:8.2f} ms", flush=True)
    g.set_option("share_local_panels", 0)
    dt, r = timeit(g); print(f"N={n} {w} virtual ranks, device copies into receive buffers: {dt*1e3:8.2f} ms", flush=True)
    g.set_option("replicate_factor", 1)
    dt, r = timeit(g, fit=True); print(f"N={n}   ... fit (replicates L x{w}):  {dt*1e3:8.2f} ms", flush=True)
    g.close()

# the serial owner path with the chip to itself: an evaluation with look-ahead off and uniform 512-wide panels runs the
# panel kernels (potrf128, panel solves, in-panel GEMMs) alone -- their summed time / 64 steps is what one owner spends per
# step on the chain of the sharded schedule (plus LA, which look-ahead-off folds into the trailing update)
h = _lib.Handle(X, y, "se_ard")
h.set_option("lookahead", 0); h.set_option("panel_wide", 0); h.set_option("dataflow_tail", 0)
h.loglik(th)
h.set_option("profile", 2); h.reset_profile(); h.loglik(th)
pr = h.profile()
steps = (n // 128 + 3) // 4
tot = pr["potrf"]["ms"] + pr["trsm"]["ms"] + pr["gemm_panel"]["ms"]
print(f"owner path, chip to itself: potrf {pr['potrf']['ms']:.2f} + panel solves {pr['trsm']['ms']:.2f} + in-panel GEMMs {pr['gemm_panel']['ms']:.2f} ms"
      f" = {tot:.2f} ms over {steps} steps = {tot/steps*1e3:.0f} us per 512-column step (trailing SYRK {pr['syrk_trailing']['ms']:.1f} ms)", flush=True)
h.close()

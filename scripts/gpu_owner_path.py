"""The SERIAL owner path of the sharded schedule with the chip to itself, step by step through the exported gphip_dist_* calls
(world = 1: this rank owns every panel; device-wide synchronisation around every step, so nothing overlaps): per outer panel the
factorisation of the panel (three launches per tile column, or ONE dataflow launch with option dist_panel_df) and the
look-ahead update of the next panel.  What an owner of an 8-GPU job spends on the chain per evaluation, minus the broadcasts.
   python scripts/gpu_owner_path.py [N [panel]]      (panel = outer panel width in 128-tiles, default the library's 4)
Writes gpurun_out/owner_path_<N>.json: per mode the per-panel step times (us, host clock around a device synchronisation,
minus the measured cost of an empty synchronisation) -- the input of scripts/scale_model.py."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesianinference_amd import _lib, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y = syn.make_dataset(n, 8)
th = syn.default_theta("se_ard", 8)
h = _lib.Handle(X, y, "se_ard")
if len(sys.argv) > 2:
    h.set_option("panel", int(sys.argv[2]))
for a in sys.argv[3:]:
    k, v = a.split("=")
    h.set_option(k, int(v))
ref = _lib.Handle(X, y, "se_ard").loglik_parts(th)
sync = torch.cuda.synchronize
sync()
t0 = time.perf_counter()
for _ in range(200):
    sync()
sync_us = (time.perf_counter() - t0) / 200 * 1e6
dump = {"N": n, "panel_tiles": int(h.get_option("panel")), "sync_us": sync_us, "modes": {}}
for df, fuse in ((0, 1), (0, 0), (1, 0), (2, 0)):
    h.set_option("dist_panel_df", df)
    h.set_option("fuse_potrf", fuse)
    for rep in range(2):
        h.set_option("profile", 2 if rep else 0)          # second pass: every launch between its own pair of events
        if rep:
            h.reset_profile()
        h.dist_begin(th, 0, 1)
        nouter = h.dist_num_panels()
        rows, _ = h.dist_panel_shape(0)
        buf = torch.empty(rows * 8, dtype=torch.uint8, device="cuda")
        sync()
        tf = tl = tr = 0.0
        per, per_la, per_rest = [], [], []
        for k in range(nouter):
            t0 = time.perf_counter(); h.dist_factor_panel(k, buf); sync(); tf += time.perf_counter() - t0
            per.append((time.perf_counter() - t0) * 1e6)
            if k + 1 < nouter:
                t0 = time.perf_counter(); h.dist_update(k, buf, k + 1, k + 2, True); sync(); tl += time.perf_counter() - t0
                per_la.append((time.perf_counter() - t0) * 1e6)
            t0 = time.perf_counter(); h.dist_update(k, buf, k + 2 if k + 1 < nouter else k + 1, nouter + 1, False); sync(); tr += time.perf_counter() - t0
            per_rest.append((time.perf_counter() - t0) * 1e6)
        ld, qd, info = h.dist_end()
        if rep == 0:                                      # (un-profiled pass: the per-step times the model uses)
            dump["modes"][f"df{df}_fuse{fuse}"] = {"factor_us": [max(v - sync_us, 0.0) for v in per], "la_us": [max(v - sync_us, 0.0) for v in per_la],
                                                   "rest_us": [max(v - sync_us, 0.0) for v in per_rest]}
    pr = h.profile()
    h.set_option("profile", 0)
    if os.environ.get("OWNER_PATH_PER_PANEL"):
        print("   us per panel factorisation (host clock, incl. one synchronisation): " + " ".join(f"{v:.0f}" for v in per[::4]), flush=True)
    ev = pr["potrf"]["ms"] + pr["trsm"]["ms"] + pr["gemm_panel"]["ms"]
    print(f"   event-timed kernels of the chain (no host time): {ev:6.2f} ms   " +
          " ".join(f"{k}={v['ms']:.2f}/{int(v['launches'])}" for k, v in pr.items() if v["launches"]), flush=True)
    print(f"N={n} panel={h.get_option('panel')} dist_panel_df={df} fuse_potrf={fuse}: panel factorisations {tf*1e3:7.2f} ms + look-ahead updates {tl*1e3:6.2f} ms = owner chain "
          f"{(tf+tl)*1e3:7.2f} ms over {nouter} panels ({(tf+tl)/nouter*1e6:.0f} us per panel); trailing updates {tr*1e3:7.2f} ms; "
          f"logdet diff {abs(ld - ref[1]) / abs(ref[1]):.1e} quad diff {abs(qd - ref[2]) / abs(ref[2]):.1e} info {info}", flush=True)
h.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"owner_path_{n}.json"), "w") as f:
    json.dump(dump, f)

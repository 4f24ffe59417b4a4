"""Does dist_panel_df = 3 hand panel columns to the broadcast stream BEFORE the owner's dataflow launch ends?
   python3 scripts/gpu_colsig_overlap.py run MODE [N]      4 virtual ranks on one GPU, panels copied into receive buffers
   python3 scripts/gpu_colsig_overlap.py analyze DIR       DIR = rocprofv3 --kernel-trace --memory-copy-trace output (csv)
The analysis counts, per chol_dataflow_kernel instance of the sharded schedule, the device-to-device panel copies that
START inside the kernel's [start, end] interval.  Mode 2 waits for the launch's event, so none can; mode 3 waits on the
column counter the launch bumps (hipStreamWaitValue32), so the copies of the early columns do."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(mode, n):
    from bayesianinference_amd import _lib, synthetic as syn
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    g = _lib.Handle(X, y, "se_ard", device=[0] * 4)
    g.set_option("shard_min_n", 0); g.set_option("share_local_panels", 0); g.set_option("dist_panel_df", mode)
    for _ in range(2):
        r = g.loglik(th)
    print(f"mode {mode} N={n}: ll={r[0]:.12g}", flush=True)
    g.close()

def analyze(d):
    kern = [f for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)]
    cop = [f for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)]
    ks, blits = [], []
    for f in kern:
        for row in csv.DictReader(open(f)):
            if "chol_dataflow_kernel" in row["Kernel_Name"]:
                ks.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
            elif "copyBuffer" in row["Kernel_Name"]:      # a same-device hipMemcpyAsync runs as the runtime's blit kernel
                blits.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    cs = []
    for f in cop:
        for row in csv.DictReader(open(f)):
            if "DEVICE_TO_DEVICE" in row.get("Direction", "").upper():
                cs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    cs += blits
    ks.sort(); cs.sort()
    inside = [sum(1 for c in cs if k[0] < c[0] < k[1]) for k in ks]
    first = [min(((c[0] - k[0]) / (k[1] - k[0]) for c in cs if k[0] < c[0] < k[1]), default=None) for k in ks]
    dur = [(k[1] - k[0]) / 1e3 for k in ks]
    n_with = sum(1 for i in inside if i)
    fr = [f for f in first if f is not None]
    print(f"{len(ks)} dataflow launches (mean {sum(dur) / max(len(dur), 1):.0f} us), {len(cs)} device-to-device copies ({len(blits)} as blit kernels); "
          f"{n_with} launches had copies starting inside them, {sum(inside)} such copies in total"
          + (f"; the first one started {100 * sum(fr) / len(fr):.0f}% of the way through its launch on average" if fr else ""))

if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 16384)
    else:
        analyze(sys.argv[2])

"""Companion of gpu_cu_partition.py: where does the time of a CU-partitioned sharded evaluation go?
   python3 scripts/gpu_cu_partition_trace.py run W [N]            (under rocprofv3 --kernel-trace; W virtual ranks on 256 / W CUs each)
   python3 scripts/gpu_cu_partition_trace.py analyze DIR          owner chain = the dataflow panel launches of the LAST evaluation:
                                                                  their durations, the gaps between them, what else ran meanwhile"""
import csv, glob, os, sys
os.environ["GPHIP_TEST_HOOKS"] = "1"          # (the CU-partition hooks are read only with this)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(W, n):
    from bayesianinference_amd import _lib, synthetic as syn
    X, y = syn.make_dataset(n, 8)
    th = syn.default_theta("se_ard", 8)
    os.environ["GPHIP_CU_PARTITION"] = "1"
    g = _lib.Handle(X, y, "se_ard", device=[0] * W)
    g.set_option("shard_min_n", 0); g.set_option("dist_owner_yield", 1)
    for k, v in (kv.split("=") for kv in os.environ.get("PART_OPTS", "").split(",") if kv):
        g.set_option(k, int(v))
    for _ in range(2):
        r = g.loglik(th)
    print(f"W={W} N={n}: ll={r[0]:.12g}", flush=True)
    g.close()

def analyze(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"].split("(")[0].replace("void gphip::", "")))
    rows.sort()
    df = [r for r in rows if r[2].startswith("chol_dataflow_kernel")]
    half = df[len(df) // 2:]                                  # the second evaluation
    t0, t1 = half[0][0], max(r[1] for r in rows)
    dur = [(r[1] - r[0]) / 1e3 for r in half]
    gaps = [(half[i + 1][0] - half[i][1]) / 1e3 for i in range(len(half) - 1)]
    span = (half[-1][1] - half[0][0]) / 1e3
    print(f"{len(half)} panel launches of the last evaluation: first start -> last end {span / 1e3:.1f} ms (evaluation ends {(t1 - half[-1][1]) / 1e6:.1f} ms later)")
    print(f"  sum of launch durations {sum(dur) / 1e3:.1f} ms (first 8: {' '.join(f'{x:.0f}' for x in dur[:8])} us; last 8: {' '.join(f'{x:.0f}' for x in dur[-8:])} us)")
    print(f"  sum of gaps between consecutive launches {sum(gaps) / 1e3:.1f} ms (first 8: {' '.join(f'{x:.0f}' for x in gaps[:8])} us; last 8: {' '.join(f'{x:.0f}' for x in gaps[-8:])} us)")
    busy = {}
    for s, e, name in rows:
        if s >= t0:
            busy[name] = busy.get(name, 0) + (e - s) / 1e6
    for name, ms in sorted(busy.items(), key=lambda kv: -kv[1])[:6]:
        print(f"  summed kernel time since the first panel launch: {name[:70]:70s} {ms:9.1f} ms")

if __name__ == "__main__" and sys.argv[1] != "queues":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 32768)
    else:
        analyze(sys.argv[2])


def queues(d):
    """per hardware queue (= stream of a rank): busy time inside the last evaluation, kernel mix, and for each panel launch what the
    same rank's other queues were doing in the 50 us before it started"""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"].split("(")[0].replace("void gphip::", "").split("<")[0],
                         row.get("Queue_Id", "?")))
    rows.sort()
    df = [r for r in rows if r[2].startswith("chol_dataflow_kernel")]
    half = df[len(df) // 2:]
    t0, t1 = half[0][0], max(r[1] for r in rows)
    per = {}
    for s, e, name, q in rows:
        if s >= t0 and not name.startswith("__amd_rocclr_streamOpsWait"):
            per.setdefault(q, {}).setdefault(name, [0, 0.0])
            per[q][name][0] += 1
            per[q][name][1] += (e - s) / 1e6
    print(f"window {(t1 - t0) / 1e6:.1f} ms; per queue: busy ms (launches) by kernel")
    for q, mix in sorted(per.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        tot = sum(v[1] for v in mix.values())
        print(f"  queue {q:>4s}: {tot:7.1f} ms busy = {100 * tot / ((t1 - t0) / 1e6):3.0f} %   " + ", ".join(f"{n} {v[1]:.1f} ({v[0]})" for n, v in sorted(mix.items(), key=lambda kv: -kv[1][1])[:3]))
    # the trailing GEMMs of each main-stream queue, step by step: (start ms, duration ms) of the 14 longest-running early ones
    for q, mix in sorted(per.items()):
        if "gemm_nt_kernel" in mix and mix["gemm_nt_kernel"][1] > 50:
            gl = [(s, e) for s, e, name, qq in rows if qq == q and s >= t0 and name == "gemm_nt_kernel" and e - s > 1e6][:14]
            print(f"  queue {q} GEMMs > 1 ms: " + " ".join(f"{(s - t0) / 1e6:.0f}+{(e - s) / 1e6:.1f}" for s, e in gl))
    # what delayed each panel launch: the latest kernel (any queue) that ENDED within 30 us before the launch started
    print("panel launch i: start (ms since first) | gap since previous panel launch's end | the kernel whose end released it")
    for i, (s, e, name, q) in enumerate(half):
        prev_end = half[i - 1][1] if i else s
        cands = [r for r in rows if s - 30000 <= r[1] <= s and r[0] < s and not r[2].startswith("__amd")]
        last = max(cands, key=lambda r: r[1]) if cands else None
        if i < 24 or i % 8 == 0:
            print(f"  {i:2d} q{q}: {(s - t0) / 1e6:7.2f} | {(s - prev_end) / 1e3:8.0f} us | " +
                  (f"{last[2]} on q{last[3]}, ran {(last[1] - last[0]) / 1e3:.0f} us, ended {(s - last[1]) / 1e3:.0f} us before" if last else "nothing ended in the 30 us before"))


if __name__ == "__main__" and sys.argv[1] == "queues":
    queues(sys.argv[2])

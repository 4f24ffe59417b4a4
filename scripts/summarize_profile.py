"""Condenses a gpurun_out/<tag>/ rocprofv3 capture (scripts/profile_gpu.sh) into the small text
summaries committed under profiles/: the --kernel-trace --stats table and per-kernel PMC averages
(HBM bytes per launch with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md §HBM).
   python scripts/summarize_profile.py <capture dir> <dst dir> <tag> [trace subdir [comma-separated PMC subdirs]]
   e.g.  ... gpurun_out/r05 profiles r05                      (trace/, pmc_fetch, pmc_write, pmc_sq, pmc_valu)
         ... gpurun_out/r05 profiles r05_cfg4 trace_cfg4 pmc_cfg4_fetch,pmc_cfg4_write,pmc_cfg4_sq"""
import collections
import csv
import os
import shutil
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
trace = sys.argv[4] if len(sys.argv) > 4 else "trace"
subs = sys.argv[5].split(",") if len(sys.argv) > 5 else ["pmc_fetch", "pmc_write", "pmc_sq", "pmc_valu"]
os.makedirs(dst, exist_ok=True)
if os.path.exists(os.path.join(src, trace, "bench_kernel_stats.csv")):
    shutil.copy(os.path.join(src, trace, "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
lines = []
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in subs:
    path = os.path.join(src, sub, "bench_counter_collection.csv")
    if not os.path.exists(path):
        continue
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines.append("kernel,counter,launches,avg_per_launch,sum")
for k in sorted(agg):
    if "rocclr" in k:
        continue
    for c, v in sorted(agg[k].items()):
        lines.append(f"{k},{c},{len(v)},{sum(v)/len(v):.6g},{sum(v):.6g}")
lines.append("")
lines.append("# derived (per launch): FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 64 B per")
lines.append("# 128-B request for wide coalesced reads -> read bytes = FETCH_SIZE*1024*2 (MI355X_MICROARCH.md).")
for k in sorted(agg):
    f_ = agg[k].get("FETCH_SIZE")
    w_ = agg[k].get("WRITE_SIZE")
    if f_ and w_:
        rd = sum(f_) / len(f_) * 1024 * 2
        wr = sum(w_) / len(w_) * 1024
        lines.append(f"# {k}: HBM-side read {rd/1e9:.4f} GB + write {wr/1e9:.4f} GB = {(rd+wr)/1e9:.4f} GB per launch")
    m = agg[k].get("SQ_VALU_MFMA_BUSY_CYCLES")
    b = agg[k].get("SQ_BUSY_CYCLES")
    if m and b and sum(m) > 0:
        lines.append(f"# {k}: SQ_VALU_MFMA_BUSY_CYCLES/launch {sum(m)/len(m):.4g}, SQ_BUSY_CYCLES/launch {sum(b)/len(b):.4g}")
    iv, im = agg[k].get("SQ_INSTS_VALU"), agg[k].get("SQ_INSTS_MFMA")
    if iv and "kbuild" in k:
        lines.append(f"# {k}: SQ_INSTS_VALU/launch {sum(iv)/len(iv):.4g}" + (f", SQ_INSTS_MFMA/launch {sum(im)/len(im):.4g}" if im else ""))
# shader clock per kernel: SQ_BUSY_CYCLES is summed over the 32 shader engines -> clock = value / 32 / duration; taken from every
# PMC pass that carries the counter together with the dispatch's timestamps (the file's launches run one at a time)
clk = collections.defaultdict(list)
for sub in subs:
    path = os.path.join(src, sub, "bench_counter_collection.csv")
    if not os.path.exists(path):
        continue
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == "SQ_BUSY_CYCLES" and "Start_Timestamp" in row and "End_Timestamp" in row:
                dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                if dur > 0:
                    name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                    clk[name].append((float(row["Counter_Value"]) / 32.0 / dur, dur))
for k in sorted(clk):
    if "rocclr" in k or not clk[k]:
        continue
    ghz = sum(c for c, _ in clk[k]) / len(clk[k])
    ms = sum(d_ for _, d_ in clk[k]) / len(clk[k]) / 1e6
    lines.append(f"# shader clock {k}: {ghz:.3f} GHz over {len(clk[k])} launches (SQ_BUSY_CYCLES / 32 / duration; average duration in that PMC pass {ms:.4f} ms)")
# what the bench printed in the SAME traced run (HIP-event averages on the library's own streams), for the recomputation of its fractions
import json
log = os.path.join(src, trace + ".log")
if os.path.exists(log):
    for ln in open(log):
        ln = ln.strip()
        if ln.startswith("{") and '"roofline"' in ln:
            try:
                rec = json.loads(ln)
            except Exception:
                continue
            r, kb = rec.get("roofline", {}), rec.get("roofline_kbuild", {})
            lines.append(f"# bench line of the traced run ({trace}.log): ms_per_step {rec.get('ms_per_step'):.3f}; trailing SYRK HIP-event average "
                         f"{r.get('avg_launch_ms'):.4f} ms over {r.get('launches')} launches = {r.get('achieved'):.2f} TFLOP/s = {r.get('frac'):.3f} of {r.get('peak')}; "
                         f"kernel build {kb.get('avg_launch_ms', float('nan')):.4f} ms = {kb.get('achieved', float('nan')):.0f} GB/s = {kb.get('frac', float('nan')):.3f}")
with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines[-12:]))

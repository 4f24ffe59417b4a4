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
with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines[-12:]))

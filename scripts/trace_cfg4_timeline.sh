#!/bin/bash
# Timeline of ONE cfg-4 batch (200 thetas x N=4096): every kernel of the last batch with start offset, duration and stream,
# then the union-busy time and the time during which no trailing SYRK (gemm_nt role 0) is running.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cfg4_timeline; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o s -- python3 $R/scripts/${SCRIPT:-gpu_batch_once.py} "$@" > $OUT/t.log 2>&1
python3 - <<PY > $OUT/timeline.txt
import csv, re
rows=[]
with open("$OUT/t/s_kernel_trace.csv") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"], r.get("Grid_Size_X","?"), r.get("Grid_Size_Y","?")))
rows.sort()
idx=max(i for i,r in enumerate(rows) if "kbuild" in r[2])
ev=rows[idx:]; t0=ev[0][0]
def short(n):
    m=re.search(r"gemm_nt_kernel<\w+, (\d), (\d)", n)
    if m: return "gemm role %s (%sx%s waves)"%(m.group(1), m.group(2), m.group(2))
    m=re.search(r"gphip::(\w+)", n)
    return m.group(1) if m else n[:30]
for s,e,n,st,gx,gy in ev:
    print("%9.1f us  +%8.1f us  stream %s  grid %s x %s  %s"%((s-t0)/1e3,(e-s)/1e3,st,gx,gy,short(n)))
span=(max(e for _,e,*_ in ev)-t0)/1e3
syrk=sorted((s,e) for s,e,n,*_ in ev if "gemm_nt_kernel<double, 0" in n or "gemm_nt_kernel<float, 0" in n)
def union(iv):
    tot=0; cs,ce=iv[0]
    for s,e in iv[1:]:
        if s<=ce: ce=max(ce,e)
        else: tot+=ce-cs; cs,ce=s,e
    return (tot+ce-cs)/1e3
print("span %.1f us; union busy %.1f us; trailing-SYRK union %.1f us (no SYRK running: %.1f us)"%(span,union(sorted((s,e) for s,e,*_ in ev)),union(syrk),span-union(syrk)))
PY
tail -3 $OUT/t.log
rm -rf $OUT/t

#!/bin/bash
# Where does a mid-size evaluation wait?  Kernel trace of the last of four evaluations at N ($1, default 16384): per stream,
# busy time vs the idle gaps between consecutive kernels (launch / dependency latency) -- the headroom a hipGraph or a
# fused schedule could recover.
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-16384}
OUT=$R/gpurun_out/trace_gaps; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o s -- python3 $R/scripts/gpu_trace_n.py $N 1 > $OUT/t.log 2>&1
python3 - <<PY
import csv, collections
rows=[]
with open("$OUT/t/s_kernel_trace.csv") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]))
rows.sort()
idx=max(i for i,r in enumerate(rows) if "kbuild" in r[2])
ev=rows[idx:]; t0=ev[0][0]; span=(max(e for _,e,_,_ in ev)-t0)/1e3
print("N=$N: %d kernels, span %.1f us"%(len(ev),span))
by=collections.defaultdict(list)
for s,e,nm,st in ev: by[st].append((s,e,nm))
for st,lst in by.items():
    busy=sum(e-s for s,e,_ in lst)/1e3
    gaps=[(lst[i+1][0]-lst[i][1])/1e3 for i in range(len(lst)-1)]
    small=[g for g in gaps if 0<=g<60]
    print("  stream %s: %d kernels, busy %.1f us, gaps<60us: n=%d sum %.1f us median %.1f us; larger gaps sum %.1f us"%(st,len(lst),busy,len(small),sum(small),sorted(small)[len(small)//2] if small else 0,sum(g for g in gaps if g>=60)))
# union busy over all streams
iv=sorted((s,e) for s,e,_,_ in ev); cur_s,cur_e=iv[0]; tot=0
for s,e in iv[1:]:
    if s<=cur_e: cur_e=max(cur_e,e)
    else: tot+=cur_e-cur_s; cur_s,cur_e=s,e
tot+=cur_e-cur_s
print("  GPU busy (union) %.1f us = %.1f %% of the span; idle %.1f us"%(tot/1e3,100*tot/1e3/span,span-tot/1e3))
PY
rm -rf $OUT/t

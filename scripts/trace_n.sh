#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-8192}
OUT=$R/gpurun_out/trace_n; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for LA in 1 0; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/la$LA -o s -- python3 $R/scripts/gpu_trace_n.py $N $LA $2 > $OUT/la$LA.log 2>&1
done
python3 - <<PY
import csv
for la in (1,0):
    rows=[]
    with open("$OUT/la%d/s_kernel_trace.csv"%la) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"], r["Queue_Id"]))
    rows.sort()
    # last evaluation = after the last kbuild kernel
    idx=max(i for i,r in enumerate(rows) if "kbuild" in r[2])
    ev=rows[idx:]
    t0=ev[0][0]
    print("lookahead=%d: %d kernels, span %.2f ms, streams %s queues %s"%(la,len(ev),(ev[-1][1]-t0)/1e6,sorted(set(r[3] for r in ev)),sorted(set(r[4] for r in ev))))
    busy=sum(e-s for s,e,*_ in ev); print("   sum durations %.2f ms"%(busy/1e6))
    for s,e,nm,st,q in ev[:28]:
        short=nm.split("<")[0].split("::")[-1]+("<"+nm.split("<")[1].split(">")[0]+">" if "<" in nm else "")
        print("   %8.1f -> %8.1f us  (%6.1f)  stream %s  %s"%((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,st,short[:40]))
PY
rm -rf $OUT/la0 $OUT/la1

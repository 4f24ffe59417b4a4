#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/small
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for LG in 0 1; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$LG -o s -- python3 $R/scripts/gpu_small_trace.py $LG > $OUT/t$LG.log 2>&1
  echo "== latency_gemm=$LG"; cut -d, -f1-4,6,7 $OUT/t$LG/s_kernel_stats.csv | head -8
done
python3 - <<PY
import csv
for lg in (0,1):
    rows=[]
    with open("$OUT/t%d/s_kernel_trace.csv"%lg) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
    rows.sort()
    # last evaluation: take the last 110 kernels; print gaps/durations of a middle stretch
    tail=rows[-110:]
    busy=sum(e-s for s,e,_ in tail); span=tail[-1][1]-tail[0][0]
    print("lg=%d last eval: span %.1f us, sum of kernel durations %.1f us, kernels %d"%(lg, span/1e3, busy/1e3, len(tail)))
    for i in range(40,52):
        s,e,nm=tail[i]
        print("   %-40s dur %6.1f us  gap_before %5.1f us"%(nm,(e-s)/1e3,(s-tail[i-1][1])/1e3))
PY
find $OUT -name "*trace.csv" -size +4M -delete

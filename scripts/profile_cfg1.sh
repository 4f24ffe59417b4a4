#!/bin/bash
# rocprofv3 kernel stats of the latency path: 300 single-theta evaluations at N=512 d=1 (cfg 1) and 40 at
# N=8192 d=8 (cfg 2).  Runs on the GPU box via gpurun; output under gpurun_out/<tag>/.
TAG=${1:-cfg1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o lat -- python3 $R/scripts/gpu_sizes.py 512 512 512 512 512 512 512 8192 8192 8192 8192 > $OUT/trace.log 2>&1
echo "trace rc=$?"
find $OUT -name "*kernel_trace.csv" -size +8M -delete
cat $OUT/trace/lat_kernel_stats.csv | head -12

# rocprofv3 kernel trace of the CU-partitioned sharded evaluation (scripts/gpu_cu_partition_trace.py), W = 2, 4, 8
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cupart; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in ${1:-2 4 8}; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/w$W -o t -- python3 $R/scripts/gpu_cu_partition_trace.py run $W 32768 > $OUT/w$W.log 2>&1
  echo "W=$W rc=$?"; grep "^W=" $OUT/w$W.log
  python3 $R/scripts/gpu_cu_partition_trace.py analyze $OUT/w$W | tee $OUT/w$W.txt
  python3 $R/scripts/gpu_cu_partition_trace.py queues $OUT/w$W > $OUT/w${W}_queues.txt
  rm -rf $OUT/w$W
done

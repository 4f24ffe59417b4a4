R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/colsig; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in 2 3; do
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/m$m -o t -- python3 $R/scripts/gpu_colsig_overlap.py run $m 16384 > $OUT/m$m.log 2>&1
  echo "mode $m rc=$?"; tail -1 $OUT/m$m.log
  python3 $R/scripts/gpu_colsig_overlap.py analyze $OUT/m$m | tee $OUT/m$m.txt

  rm -rf $OUT/m$m
done

#!/bin/bash
# cfg 4 (200 thetas x N=4096, one batched call): rocprofv3 kernel stats + HBM / SQ PMC passes, each in its own run.
TAG=${1:-r02_cfg4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/scripts/gpu_batch_once.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ARGS > $OUT/trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 $ARGS > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o bench -- python3 $ARGS > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT

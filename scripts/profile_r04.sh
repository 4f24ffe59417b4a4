#!/bin/bash
# Round-4 capture (runs on the GPU box via gpurun): rocprofv3 kernel stats of the default bench command, the HBM / SQ PMC
# passes (each in its own run, never combined with a trace), kernel stats of the cfg-5 fp32 leg, VALU / HBM PMC passes of the
# fp32 Matern kernel build (its round-4 4-rows-per-lane packed path) and of the fp64 build.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-alone"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 $ARGS > $OUT/pmc_write.log 2>&1
echo "write rc=$?"
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o bench -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
echo "sq rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg5 -o cfg5 -- python3 $R/scripts/gpu_cfg5.py > $OUT/trace_cfg5.log 2>&1
echo "cfg5 rc=$?"
K32="$R/scripts/gpu_kbuild_f32.py"
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_k32_sq -o k32 -- python3 $K32 > $OUT/pmc_k32_sq.log 2>&1
echo "k32 sq rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_k32_fetch -o k32 -- python3 $K32 > $OUT/pmc_k32_fetch.log 2>&1
echo "k32 fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_k32_write -o k32 -- python3 $K32 > $OUT/pmc_k32_write.log 2>&1
echo "k32 write rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k32 -o k32 -- python3 $K32 > $OUT/trace_k32.log 2>&1
echo "k32 trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_k64_valu -o bench -- python3 $ARGS > $OUT/pmc_k64_valu.log 2>&1
echo "k64 valu rc=$?"
find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT

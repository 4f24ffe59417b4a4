#!/bin/bash
# Round-5 capture (runs on the GPU box via gpurun): rocprofv3 kernel stats of the default bench command, the HBM / SQ PMC
# passes (each in its own run, never combined with a trace), the kernel build's VALU / MFMA counters (fp64 SE-ARD d=8 inside
# the bench command; fp32 Matern d=16 in scripts/gpu_kbuild_f32.py), kernel stats of the cfg-5 fp32 leg, and kernel stats +
# PMC passes of the cfg-4 batch (200 x N=4096).   bash scripts/profile_r05.sh [tag]   then scripts/summarize_profile.py
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-alone"
pass() {   # name, counters.. -- program args..
    local name=$1; shift
    local ctr=()
    while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
    shift
    timeout 300 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $OUT/$name -o bench -- python3 "$@" > $OUT/$name.log 2>&1
    echo "$name rc=$?"
}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?"
pass pmc_fetch FETCH_SIZE -- $ARGS
pass pmc_write WRITE_SIZE -- $ARGS
pass pmc_sq SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $ARGS
pass pmc_valu SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES -- $ARGS
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg5 -o bench -- python3 $R/scripts/gpu_cfg5.py > $OUT/trace_cfg5.log 2>&1
echo "cfg5 trace rc=$?"
K32="$R/scripts/gpu_kbuild_f32.py"
pass pmc_k32_sq SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA -- $K32
pass pmc_k32_fetch FETCH_SIZE -- $K32
pass pmc_k32_write WRITE_SIZE -- $K32
C4="$R/scripts/gpu_batch_once.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg4 -o bench -- python3 $C4 > $OUT/trace_cfg4.log 2>&1
echo "cfg4 trace rc=$?"
pass pmc_cfg4_fetch FETCH_SIZE -- $C4
pass pmc_cfg4_write WRITE_SIZE -- $C4
pass pmc_cfg4_sq SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $C4
find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT

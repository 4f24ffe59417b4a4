#!/bin/bash
# Round-6 capture (runs on the GPU box via gpurun), AFTER the last kernel commit: rocprofv3 kernel stats of the default bench
# command, the HBM / SQ PMC passes (each in its own run, never combined with a trace), kernel stats of the cfg-5 fp32 leg,
# kernel stats + PMC passes of the cfg-4 batch, and -- new -- kernel stats + FETCH_SIZE of the single-launch evaluation at
# N = 8192 (cfg 2).  scripts/summarize_profile.py writes the shader clock of every kernel (SQ_BUSY_CYCLES / 32 / duration
# from the pmc_sq pass) and the HIP-event averages the SAME traced run printed into the summary, so that the bench line's
# fractions can be recomputed from profiles/ alone.   bash scripts/profile_r06.sh [tag]
TAG=${1:-r06}
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
C4="$R/scripts/gpu_batch_once.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg4 -o bench -- python3 $C4 > $OUT/trace_cfg4.log 2>&1
echo "cfg4 trace rc=$?"
pass pmc_cfg4_fetch FETCH_SIZE -- $C4
pass pmc_cfg4_write WRITE_SIZE -- $C4
pass pmc_cfg4_sq SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $C4
C2="$R/scripts/gpu_latency.py 8192"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg2 -o bench -- python3 $C2 > $OUT/trace_cfg2.log 2>&1
echo "cfg2 trace rc=$?"
pass pmc_cfg2_fetch FETCH_SIZE -- $C2
pass pmc_cfg2_write WRITE_SIZE -- $C2
pass pmc_cfg2_sq SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -- $C2
find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT

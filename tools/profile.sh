#!/bin/bash
# GPU-box profile of the bench command: kernel trace + stats, then HBM traffic counters in two
# separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Output: gpurun_out/prof_$TAG/
# usage: [PROFILE_EXTRA="..."] bash tools/profile.sh TAG        (summarise afterwards with tools/summarize_profile.py)
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export NR_SIDE_STREAM=0      # per-kernel durations: one kernel at a time (bench.py's roofline leg does the same)
ARGS="bench.py --steps 5 --warmup 2 --repeats 0 --no-cpu-baseline --no-eval --no-sweep $PROFILE_EXTRA"     # PROFILE_EXTRA: e.g. "--workload bert --gemm bf16 --batch-size 8"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ARGS > $OUT/pmc_mfma.log 2>&1
echo profile done

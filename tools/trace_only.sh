#!/bin/bash
# kernel trace of a few bench steps (one kernel at a time) + the per-dispatch view of the last step.  usage: bash tools/trace_only.sh TAG
set -e
TAG=${1:-t}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export NR_SIDE_STREAM=0
ARGS="bench.py --steps 5 --warmup 2 --repeats 0 --no-cpu-baseline --no-eval --no-sweep $PROFILE_EXTRA"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
python3 tools/trace_step.py $TAG > $OUT/step.txt
tail -3 $OUT/step.txt

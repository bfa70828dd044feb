#!/bin/bash
# host floor + headline bench on the GPU box; results under gpurun_out/r03/ with the given tag
set -o pipefail
TAG=${1:-x}
mkdir -p gpurun_out/r03
timeout -k 10 120 python tools/host_floor.py 2 > gpurun_out/r03/host_floor_$TAG.txt 2>&1 || exit 1
timeout -k 10 400 python bench.py --no-cpu-baseline --no-eval > gpurun_out/r03/bench_$TAG.json 2> gpurun_out/r03/bench_$TAG.err || exit 1
cat gpurun_out/r03/host_floor_$TAG.txt
python - <<PY
import json
d = json.load(open("gpurun_out/r03/bench_$TAG.json"))
print(d["value"], d["ms_per_step"], d["repeats_ms_per_step"])
print(d["kernel_ms_per_step"])
print("x3 frac", d["roofline"]["frac"], "avg launch ms", d["roofline"]["avg_launch_ms"], "other", d["roofline"]["other_gemm_launches"]["ms_per_step"])
print(d["hbm_families_in_step"])
PY

#!/bin/bash
# rocprofv3 --kernel-trace --stats over the bench step -> gpurun_out/<tag>_kernel_stats.csv (+ the bench line under the profiler)
#   usage (GPU box, repo root): bash tools/prof_bench.sh tag [bench args]
REPO=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
cd /tmp && rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python "$REPO/bench.py" --no-cpu-baseline --no-hbm-loop "$@" > "$REPO/gpurun_out/${TAG}_bench_under_profiler.log" 2>&1
cp "$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)" "$REPO/gpurun_out/${TAG}_kernel_stats.csv"

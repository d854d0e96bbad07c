#!/bin/bash
# rocprofv3 --kernel-trace of the bench step (every dispatch with its timestamps) -> gpurun_out/<tag>_gaps.txt (tools/gap_report.py)
#   usage (GPU box, repo root): bash tools/trace_bench.sh tag [bench args]
REPO=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
cd /tmp && rm -rf /tmp/trace_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$TAG -o p -- python "$REPO/bench.py" --no-cpu-baseline --no-hbm-loop "$@" > "$REPO/gpurun_out/${TAG}_trace_bench.log" 2>&1
F=$(find /tmp/trace_$TAG -name '*kernel_trace.csv' | head -1)
python "$REPO/tools/gap_report.py" "$F" > "$REPO/gpurun_out/${TAG}_gaps.txt" 2>&1
python "$REPO/tools/stream_report.py" "$F" --dump "$REPO/gpurun_out/${TAG}_steps.csv.gz" > "$REPO/gpurun_out/${TAG}_streams.txt" 2>&1

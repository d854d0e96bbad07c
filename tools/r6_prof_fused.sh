#!/bin/bash
mkdir -p gpurun_out/r6c
bash tools/prof_bench.sh r6c/fused --steps 8 --warmup 3
bash tools/trace_bench.sh r6c/fused --steps 8 --warmup 3
head -24 gpurun_out/r6c/fused_streams.txt

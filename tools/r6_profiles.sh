#!/bin/bash
# round-6 profile set of the bench command on one box: kernel stats + stream critical path of the product configuration, a
# SERIALIZED kernel trace (weight gradients and mask head on the main stream: per-kernel averages reproducible from profiles/),
# whole-step HBM-side traffic (two PMC passes), the layer table
mkdir -p gpurun_out/r6f
bash tools/prof_bench.sh r6f/final --steps 8 --warmup 3
bash tools/trace_bench.sh r6f/final --steps 8 --warmup 3
CFUN_WGRAD_STREAM=0 CFUN_OVERLAP_MASK_HEAD=0 bash tools/prof_bench.sh r6f/serialized --steps 8 --warmup 3
bash tools/pmc_step.sh r6 > gpurun_out/r6f/pmc_step_summary.txt 2>&1
cp gpurun_out/pmc_r6_FETCH_SIZE.csv gpurun_out/pmc_r6_WRITE_SIZE.csv gpurun_out/r6f/ 2>/dev/null
( timeout 600 python tools/bench_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6f/layers.log
head -12 gpurun_out/r6f/final_streams.txt
head -8 gpurun_out/r6f/final_gaps.txt
tail -22 gpurun_out/r6f/pmc_step_summary.txt
tail -2 gpurun_out/r6f/serialized_bench_under_profiler.log | cut -c1-200

#!/bin/bash
# The round-5 profile set of the bench command on one box: kernel stats, stream critical path, PMC passes of the dominant conv,
# whole-step HBM-side traffic.   bash tools/r5_profiles.sh   (results under gpurun_out/r5prof/)
mkdir -p gpurun_out/r5prof
bash tools/prof_bench.sh r5prof/final --steps 8 --warmup 3
bash tools/trace_bench.sh r5prof/final --steps 8 --warmup 3
bash tools/pmc_traffic.sh > gpurun_out/r5prof/pmc_wino_l4_0.json 2> gpurun_out/r5prof/pmc_traffic.err
cp gpurun_out/pmc_FETCH_SIZE.csv gpurun_out/r5prof/pmc_FETCH_SIZE_wino_l4_0.csv; cp gpurun_out/pmc_WRITE_SIZE.csv gpurun_out/r5prof/pmc_WRITE_SIZE_wino_l4_0.csv
bash tools/pmc_mfma.sh > gpurun_out/r5prof/pmc_mfma_wino_l4_0.json 2> gpurun_out/r5prof/pmc_mfma.err
bash tools/pmc_step.sh r5 > gpurun_out/r5prof/pmc_step_summary.txt 2>&1
cp gpurun_out/pmc_r5_FETCH_SIZE.csv gpurun_out/r5prof/ 2>/dev/null; cp gpurun_out/pmc_r5_WRITE_SIZE.csv gpurun_out/r5prof/ 2>/dev/null
( timeout 600 python tools/bench_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5prof/layers.log
tail -3 gpurun_out/r5prof/final_bench_under_profiler.log | cut -c1-300
head -12 gpurun_out/r5prof/final_streams.txt
tail -25 gpurun_out/r5prof/pmc_step_summary.txt
cat gpurun_out/r5prof/pmc_wino_l4_0.json | head -30

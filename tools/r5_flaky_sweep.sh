#!/bin/bash
# Repeat the stream- / rank-sensitive GPU tests to look for flakiness -> gpurun_out/r5flaky/sweep.log
mkdir -p gpurun_out/r5flaky
: > gpurun_out/r5flaky/sweep.log
for i in 1 2 3; do
  ( timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_modules_gpu.py -q -x -k "four_ranks or two_ranks_on_real or lagging or gradient_reducer_streams or allocator_churn or mask_head_side_stream or cfg2_full_size or two_models" 2>&1 | grep -E "passed|failed|error" | tail -2 ) >> gpurun_out/r5flaky/sweep.log 2>&1
done
cat gpurun_out/r5flaky/sweep.log

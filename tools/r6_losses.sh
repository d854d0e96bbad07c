#!/bin/bash
# one-pass mask losses vs the round-5 kernels at 4 x 192^3 x 8
mkdir -p gpurun_out/r6b
( python tools/bench_losses.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6b/losses.log
cat gpurun_out/r6b/losses.log
python -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -x -q -k "mask_losses or losses" 2>&1 | tail -2

#!/bin/bash
# round 6: regenerate the fp64 gradient fixture for ALL 27 U-Net conv weights (host CPU of the GPU box, ~8 min) while the GPU tier runs
mkdir -p gpurun_out/r6d
( CFUN_GEN_THREADS=48 CFUN_GEN_OUT=gpurun_out/r6d/grad_fp64_cfg2.npz timeout 2400 python tests/golden/gen_grad_fp64_cfg2.py > gpurun_out/r6d/gen_grad_fp64.log 2>&1 ) &
GEN=$!
( timeout 3000 python -m pytest tests -m gpu -q -x --deselect tests/test_modules_gpu.py::test_cfg2_full_size_step_properties 2>&1 | tail -15 ) > gpurun_out/r6d/gpu_tier.log 2>&1
wait $GEN
tail -5 gpurun_out/r6d/gen_grad_fp64.log
cat gpurun_out/r6d/gpu_tier.log

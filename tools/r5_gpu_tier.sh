#!/bin/bash
# The GPU tier with the slowest tests listed -> gpurun_out/r5tier/gpu_tier.log   (bash tools/r5_gpu_tier.sh on the GPU box)
mkdir -p gpurun_out/r5tier
( timeout 2400 python -m pytest tests -q -m gpu --durations=12 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -60 ) > gpurun_out/r5tier/gpu_tier.log 2>&1
tail -22 gpurun_out/r5tier/gpu_tier.log

mkdir -p gpurun_out/r5tier
( timeout 2400 python -m pytest tests -q -m gpu --durations=15 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -60 ) > gpurun_out/r5tier/gpu_tier.log 2>&1
tail -30 gpurun_out/r5tier/gpu_tier.log

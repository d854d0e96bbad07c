mkdir -p gpurun_out/r5c19
( timeout 900 python -m pytest tests/test_modules_gpu.py -q -k "lagging or gradient_reducer_streams" 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25 ) > gpurun_out/r5c19/t.log 2>&1
tail -25 gpurun_out/r5c19/t.log

mkdir -p gpurun_out/r5c11
( timeout 1200 python -m pytest tests/test_modules_gpu.py tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -x -q -k "roi or pyramid or resize_known or cfg2_full_size or mask_head_side_stream or gradient_reducer_streams or classifier" 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -60 ) > gpurun_out/r5c11/tests.log 2>&1
tail -60 gpurun_out/r5c11/tests.log

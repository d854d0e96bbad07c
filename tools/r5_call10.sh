mkdir -p gpurun_out/r5c10
( CFUN_GEN_THREADS=96 CFUN_GEN_OUT=gpurun_out/r5c10/grad_fp64_cfg2.npz python tests/golden/gen_grad_fp64_cfg2.py > gpurun_out/r5c10/gen_fp64.log 2>&1 ) &
GEN=$!
( timeout 1200 python -m pytest tests/test_modules_gpu.py tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -x -q -k "roi or pyramid or resize_known or cfg2_full_size or mask_head_side_stream or gradient_reducer_streams or classifier" 2>&1 | grep -v amdgpu.ids | tail -6 ) > gpurun_out/r5c10/tests.log 2>&1
tail -4 gpurun_out/r5c10/tests.log
wait $GEN
tail -45 gpurun_out/r5c10/gen_fp64.log

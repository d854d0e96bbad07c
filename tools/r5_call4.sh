mkdir -p gpurun_out/r5c4
( timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_dist_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -15 ) > gpurun_out/r5c4/modules_gpu.log 2>&1
tail -6 gpurun_out/r5c4/modules_gpu.log
bash tools/r5_ab_env.sh r5c4 "CFUN_WGRAD_STREAM=0" "CFUN_WGRAD_STREAM=1"

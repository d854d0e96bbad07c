mkdir -p gpurun_out/r5c1
( timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -40 ) > gpurun_out/r5c1/dist_gpu.log 2>&1
( timeout 300 python -m pytest tests/test_modules_gpu.py tests/test_kernels_gpu.py -x -q -k "async_scalar or pointwise" 2>&1 | tail -5 ) > gpurun_out/r5c1/small_tests.log 2>&1
( CFUN_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8 --sharded --steps 2 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -20 ) > gpurun_out/r5c1/bench_sharded_world8_gloo.log 2>&1
( timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r5c1/bench_base.log 2>&1
( timeout 600 python tools/bench_layers.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5c1/layers_base.log 2>&1
tail -3 gpurun_out/r5c1/dist_gpu.log gpurun_out/r5c1/small_tests.log; tail -c 600 gpurun_out/r5c1/bench_sharded_world8_gloo.log; tail -c 400 gpurun_out/r5c1/bench_base.log

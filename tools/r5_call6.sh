mkdir -p gpurun_out/r5c6
( CFUN_WGRAD_STREAM=0 python tools/cpu_enqueue.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5c6/enqueue_w0.log
( CFUN_WGRAD_STREAM=1 python tools/cpu_enqueue.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5c6/enqueue_w1.log
cat gpurun_out/r5c6/enqueue_w0.log gpurun_out/r5c6/enqueue_w1.log
CFUN_WGRAD_STREAM=1 bash tools/trace_bench.sh r5c6/w1 --steps 8 --warmup 3
head -50 gpurun_out/r5c6/w1_streams.txt
head -40 gpurun_out/r5c6/w1_gaps.txt

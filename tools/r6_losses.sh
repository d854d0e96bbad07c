#!/bin/bash
# one-pass mask losses vs the round-5 kernels at 4 x 192^3 x 8, both register builds of the backward
mkdir -p gpurun_out/r6b
( python tools/bench_losses.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6b/losses_cap.log

cat gpurun_out/r6b/losses_cap.log
python -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -x -q -k "mask_losses or losses" 2>&1 | tail -3
for v in 1 0; do ( CFUN_FUSED_MASK_LOSS=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-loop 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused=$v', 'value %.3f ms %.3f' % (d['value'], d['ms_per_step']), d['losses'])" ); done

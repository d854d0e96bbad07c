#!/bin/bash
mkdir -p gpurun_out/r6e
( timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 ) > gpurun_out/r6e/gpu_tier.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -2 ) > gpurun_out/r6e/bench_full.log 2>&1; echo rc=$? >> gpurun_out/r6e/bench_full.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 ) > gpurun_out/r6e/smoke.log 2>&1
tail -12 gpurun_out/r6e/gpu_tier.log; cat gpurun_out/r6e/smoke.log; tail -c 600 gpurun_out/r6e/bench_full.log

mkdir -p gpurun_out/r5final
( timeout 1500 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r5final/bench_full.log 2>&1; echo rc=$? >> gpurun_out/r5final/bench_full.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r5final/smoke.log 2>&1
tail -2 gpurun_out/r5final/smoke.log
tail -c 1500 gpurun_out/r5final/bench_full.log

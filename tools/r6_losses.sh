#!/bin/bash
# one-pass mask losses at 4 x 192^3 x 8: pair-per-thread (256 threads) vs voxel-per-thread (512 threads) builds of both kernels
mkdir -p gpurun_out/r6b
for v in "2 2" "1 1" "2 1" "1 2" "2 2" "1 1"; do set -- $v; echo "FWD_VPT=$1 BWD_VPT=$2"; ( CFUN_FUSED_FWD_VPT=$1 CFUN_FUSED_BWD_VPT=$2 python tools/bench_losses.py 2>&1 | grep -v amdgpu.ids | tail -2 ); done | tee gpurun_out/r6b/losses_vpt.log
CFUN_FUSED_FWD_VPT=1 CFUN_FUSED_BWD_VPT=1 python -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -x -q -k "mask_losses or losses" 2>&1 | tail -1

#!/bin/bash
# conv3d_c1_1 (1 -> 20, 3x3x3 @ 4 x 96^3): the 128-thread two-x-halves kernel (CFUN_STEM_XH=1) against the round-5 kernel, same box
for i in 1 2; do
for v in "CFUN_STEM_XH=0" "CFUN_STEM_XH=1" "CFUN_STEM_XH=1 CFUN_STEM_ZPT=1" "CFUN_STEM_XH=0 CFUN_STEM_ZPT=4"; do echo "$v: $(env $v python tools/bench_b2b.py stem 2>&1 | grep -v amdgpu.ids | tail -1)"; done
done
python -m pytest tests/test_kernels_gpu.py -x -q -k "stem or conv" 2>&1 | tail -2

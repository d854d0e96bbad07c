#!/bin/bash
for z in 32 16 24 48 64 96 32; do echo "CFUN_FUSED_ZS=$z: $(CFUN_FUSED_ZS=$z python tools/bench_losses.py 2>&1 | grep -v amdgpu.ids | grep 'one-pass forward')"; done

#!/bin/bash
# A/B of the one-pass mask losses inside the step, alternating order (same box)
for v in 0 1 0 1 0 1; do ( CFUN_FUSED_MASK_LOSS=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-loop 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused=$v', 'value %.3f ms %.3f' % (d['value'], d['ms_per_step']))" ); done

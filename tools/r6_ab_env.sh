#!/bin/bash
# A/B of one environment switch inside the step, alternating order (same box):  bash tools/r6_ab_env.sh VAR [reps]
V=$1; R=${2:-3}
for i in $(seq $R); do for v in 0 1; do ( env $V=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-loop 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V=$v', 'value %.3f ms %.3f' % (d['value'], d['ms_per_step']))" ); done; done

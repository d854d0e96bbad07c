#!/bin/bash
mkdir -p gpurun_out/r6g
( timeout 900 python bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-loop 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3', 'value %.3f ms %.3f' % (d['value'], d['ms_per_step']))" )
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_cfg3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg3 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-loop > /dev/null 2>&1
cp "$(find /tmp/prof_cfg3 -name '*kernel_stats.csv' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/r6g/cfg3_kernel_stats.csv
head -25 $GRAFT_REPO_ROOT/gpurun_out/r6g/cfg3_kernel_stats.csv | cut -c1-110

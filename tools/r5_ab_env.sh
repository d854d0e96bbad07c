#!/bin/bash
# A/B of environment knobs over the bench step, same box, same call:  bash tools/r5_ab_env.sh tag "VAR=1" "VAR=0" ...
# every variant runs bench.py --no-cpu-baseline --no-hbm-loop --steps 20 --warmup 5 and prints ms per step
TAG=$1; shift
mkdir -p gpurun_out/$TAG
i=0
for v in "$@"; do
  ( env $v timeout 600 python bench.py --no-cpu-baseline --no-hbm-loop --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/$TAG/bench_$i.log 2>&1
  python - "$v" gpurun_out/$TAG/bench_$i.log <<'PY'
import json, sys
ok = False
for line in open(sys.argv[2]):
    if line.startswith("{"):
        d = json.loads(line); ok = True
        print("%-40s %.3f ms/step  losses %s" % (sys.argv[1] or "<default>", d["ms_per_step"], ["%.6g" % l for l in d["losses"]]))
if not ok:
    print(sys.argv[1], "FAILED:", open(sys.argv[2]).read()[-600:])
PY
  i=$((i + 1))
done

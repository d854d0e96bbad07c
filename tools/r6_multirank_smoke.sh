#!/bin/bash
# bench.py's data-parallel line WITH its one-volume leg on 4 and 8 gloo ranks sharing one GPU (plain python: the script launches itself)
mkdir -p gpurun_out/r6m
for n in 4 8; do
  ( CFUN_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus $n --steps 2 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1 ) > gpurun_out/r6m/bench_gpus${n}_gloo.json 2>&1
  python - gpurun_out/r6m/bench_gpus${n}_gloo.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    leg = d.get("sharded_one_volume") or {}
    print(sys.argv[1], "n_gpus", d["n_gpus"], "value %.2f" % d["value"], "preflight", d["preflight"]["ok"], "leg", {k: leg.get(k) for k in ("value", "ms_per_step", "error")},
          "parity", (d.get("sharded_parity") or {}).get("rel_diff"))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[1]).read()[-1500:])
PY
done

#!/bin/bash
# round 6, first GPU call: the N-rank self-launch of bench.py (gloo on the 1-GPU box), a quick N = 1 bench, and a fresh kernel
# stats + trace of the final round-5 tree (the round-5 traces predate the fold kernels)
mkdir -p gpurun_out/r6a
( CFUN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r6a/bench_gpus2_selflaunch_gloo.log 2>&1
( timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2 ) > gpurun_out/r6a/bench_quick.log 2>&1
bash tools/prof_bench.sh r6a/base --steps 8 --warmup 3
bash tools/trace_bench.sh r6a/base --steps 8 --warmup 3
python - <<'PY'
import json
for f in ("gpurun_out/r6a/bench_gpus2_selflaunch_gloo.log", "gpurun_out/r6a/bench_quick.log"):
    ok = False
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line); ok = True
            print(f, "n_gpus", d["n_gpus"], "value %.2f" % d["value"], "ms %.2f" % d["ms_per_step"], "preflight", (d.get("preflight") or {}).get("ok"),
                  "sharded leg", {k: v for k, v in (d.get("sharded_one_volume") or {}).items() if k in ("value", "ms_per_step", "error")},
                  "parity", (d.get("sharded_parity") or {}).get("rel_diff"))
    if not ok:
        print(f, "FAILED", open(f).read()[-2000:])
PY
head -30 gpurun_out/r6a/base_gaps.txt
head -14 gpurun_out/r6a/base_streams.txt
head -40 gpurun_out/r6a/base_kernel_stats.csv | cut -c1-150

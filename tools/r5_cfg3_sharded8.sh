#!/bin/bash
# BASELINE configs[3] in its specified form -- ONE 512x512x256 volume depth-sharded over 8 ranks (32 input planes = 2 p3 planes per
# rank, 4 RoIs x 2 ranks z-sharded U-Nets) -- on ONE GPU over gloo, with bench.py's parity leg against the single-process step
mkdir -p gpurun_out/r5cfg3
( CFUN_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29588 bench.py --gpus 8 --sharded --workload cfg3 --steps 2 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -6 ) > gpurun_out/r5cfg3/bench_cfg3_sharded8_gloo.log 2>&1
python - <<'PY'
import json
ok = False
for line in open("gpurun_out/r5cfg3/bench_cfg3_sharded8_gloo.log"):
    if line.startswith("{"):
        d = json.loads(line); ok = True
        print(d["config"]["workload"][:60], "| n_gpus", d["n_gpus"], d["scaling"], "| ms/step %.0f" % d["ms_per_step"])
        print("losses (sum of shares)", d["losses"])
        print("sharded_parity", d["sharded_parity"]["rel_diff"], d["sharded_parity"]["ok"])
if not ok:
    print(open("gpurun_out/r5cfg3/bench_cfg3_sharded8_gloo.log").read()[-2500:])
PY

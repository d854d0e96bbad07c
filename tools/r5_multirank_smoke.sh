#!/bin/bash
# bench.py's N > 1 paths on ONE GPU over gloo (RCCL refuses several ranks per device): data-parallel replicas (what the driver
# scales) with 2 and 8 ranks, and ONE volume over 8 ranks (--sharded, with its parity leg) -> gpurun_out/r5multi/*.log
mkdir -p gpurun_out/r5multi
run() { tag=$1; n=$2; shift 2
  ( CFUN_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 295$n$n bench.py --gpus $n --steps 3 --warmup 1 "$@" 2>&1 | grep -v amdgpu.ids | tail -4 ) > gpurun_out/r5multi/$tag.log 2>&1
  python - gpurun_out/r5multi/$tag.log <<'PY'
import json, sys
ok = False
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); ok = True
        print(sys.argv[1], "n_gpus", d["n_gpus"], "scaling", d["scaling"], "value %.2f" % d["value"], "ms/step %.1f" % d["ms_per_step"],
              "losses", ["%.5g" % l for l in d["losses"]], "sharded_parity", (d.get("sharded_parity") or {}).get("rel_diff"))
if not ok:
    print(sys.argv[1], "FAILED:", open(sys.argv[1]).read()[-1500:])
PY
}
run dp2 2
run dp8 8
run sharded8 8 --sharded
run sharded4 4 --sharded

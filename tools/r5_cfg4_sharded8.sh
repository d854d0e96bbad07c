mkdir -p gpurun_out/r5cfg4
( CFUN_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29589 bench.py --gpus 8 --sharded --workload cfg4 --steps 2 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -12 ) > gpurun_out/r5cfg4/log.txt 2>&1
python - <<'PY'
import json
ok = False
for line in open("gpurun_out/r5cfg4/log.txt"):
    if line.startswith("{"):
        d = json.loads(line); ok = True
        print(d["config"]["workload"][:70], "| ms/step %.0f" % d["ms_per_step"]); print(d["losses"]); print(d["sharded_parity"]["rel_diff"], d["sharded_parity"]["ok"])
if not ok: print(open("gpurun_out/r5cfg4/log.txt").read()[-2500:])
PY

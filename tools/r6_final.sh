#!/bin/bash
# round 6, final tree: GPU tier, smoke, the driver's bench command, the product-configuration trace (stream critical path, gaps)
mkdir -p gpurun_out/r6h
( timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|^$\|^  " | tail -12 ) > gpurun_out/r6h/gpu_tier.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 ) > gpurun_out/r6h/smoke.log 2>&1
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 ) > gpurun_out/r6h/bench_full.json 2>gpurun_out/r6h/bench_full.err; echo rc=$? >> gpurun_out/r6h/bench_full.err
bash tools/trace_bench.sh r6h/final --steps 8 --warmup 3
( CFUN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1 ) > gpurun_out/r6h/bench_gpus2_gloo.json 2>&1
cat gpurun_out/r6h/gpu_tier.log gpurun_out/r6h/smoke.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6h/bench_full.json").read().strip().splitlines()[-1])
print("bench: value %.3f ms %.3f frac %.3f hbm %.3f p3d %s loss_parity %s grad_parity %s (%d tensors)" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_hbm"]["frac"], {k: round(v,3) for k,v in d["roofline_hbm"]["p3d_stem"].items() if k in ("avg_call_ms","hbm_frac","valu_frac")}, d["loss_parity"]["ok"], d["grad_parity"]["ok"], len(d["grad_parity"]["tensors"])))
d=json.loads(open("gpurun_out/r6h/bench_gpus2_gloo.json").read().strip().splitlines()[-1])
print("gpus2 gloo: n_gpus", d["n_gpus"], "value %.2f" % d["value"], "preflight", d["preflight"]["ok"], "leg", {k: d["sharded_one_volume"].get(k) for k in ("value","ms_per_step","error")}, "parity", d["sharded_parity"]["ok"])
PY
head -14 gpurun_out/r6h/final_streams.txt; head -6 gpurun_out/r6h/final_gaps.txt

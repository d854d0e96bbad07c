#!/bin/bash
# HBM-side traffic of the dominant conv launch (conv_norm_lrelu_l4.0: 3x3x3 40->40 @ 4x96^3) from the TCC counters,
# collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes with
# --kernel-trace only.  Writes gpurun_out/pmc_{FETCH,WRITE}_SIZE.csv and prints the JSON that is committed as
# profiles/roundN_pmc_wino_l4_0.json (and read by bench.py for roofline.traffic).
#   usage (on the GPU box, from the repo root):  bash tools/pmc_traffic.sh
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- \
      python "$REPO/tools/bench_layers.py" --filter l4.0 --iters 1 > "$REPO/gpurun_out/pmc_$c.log" 2>&1
  cp "$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)" "$REPO/gpurun_out/pmc_$c.csv"
done
python "$REPO/tools/pmc_traffic.py" "$REPO/gpurun_out/pmc_FETCH_SIZE.csv" "$REPO/gpurun_out/pmc_WRITE_SIZE.csv"

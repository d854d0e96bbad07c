#!/bin/bash
# Whole-step HBM-side traffic: FETCH_SIZE and WRITE_SIZE of EVERY kernel of the bench step in
# SEPARATE rocprofv3 --pmc passes (--kernel-trace only), summed per kernel family by tools/pmc_step.py and set against
# the ~30 GB/step algorithmic figure of SURVEY.md section 8(d).
#   usage (on the GPU box, from the repo root):  bash tools/pmc_step.sh [tag]
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-step}
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_$c -o p -- \
      python "$REPO/bench.py" --no-cpu-baseline --no-hbm-loop --steps 2 --warmup 1 > "$REPO/gpurun_out/pmc_${TAG}_$c.log" 2>&1
  f=$(find /tmp/pmcs_$c -name '*counter_collection.csv' | head -1)
  python "$REPO/tools/pmc_step.py" --compact "$f" "$REPO/gpurun_out/pmc_${TAG}_$c.csv"
done
python "$REPO/tools/pmc_step.py" --steps 3 "$REPO/gpurun_out/pmc_${TAG}_FETCH_SIZE.csv" "$REPO/gpurun_out/pmc_${TAG}_WRITE_SIZE.csv" \
    | tee "$REPO/gpurun_out/pmc_${TAG}_summary.txt"

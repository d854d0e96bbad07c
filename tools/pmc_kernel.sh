#!/bin/bash
# Issue / wait counters of the kernels one bench_layers filter launches, three rocprofv3 --pmc passes (counters only: no traces)
#   usage (GPU box, repo root): bash tools/pmc_kernel.sh tag "layer filter" kernel_regex
REPO=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; FILTER=$2; KRE=$3
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
OUT="$REPO/gpurun_out/${TAG}_pmc_kernel.txt"
: > "$OUT"
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  cd /tmp && rm -rf /tmp/pmck_$TAG_$i
  rocprofv3 --pmc $SET --output-format csv -d /tmp/pmck_${TAG}_$i -o p -- python "$REPO/tools/bench_layers.py" --filter "$FILTER" --iters 1 > /tmp/pmck_${TAG}_$i.log 2>&1
  F=$(find /tmp/pmck_${TAG}_$i -name '*counter_collection.csv' | head -1)
  python - "$F" "$KRE" >> "$OUT" <<'PY'
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    if not re.search(sys.argv[2], r["Kernel_Name"]):
        continue
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
    k = m.group(1) if m else r["Kernel_Name"][:50]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"])); n[k] += 1
for k, v in agg.items():
    print(k, "dispatches", n[k], {c: round(x / n[k], 1) for c, x in sorted(v.items())})
PY
done
cat "$OUT"

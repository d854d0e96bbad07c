#!/bin/bash
# Wave-state / LDS / cache counters of selected kernels: where do their waves spend their cycles?
#   usage (GPU box, repo root): bash tools/pmc_sq.sh "<kernel regex>" <python script + args ...>
#   e.g.  bash tools/pmc_sq.sh "k_conv_wino|k_wgrad_wino" tools/bench_layers.py --filter l4.0 --iters 1
REPO=$(cd "$(dirname "$0")/.." && pwd)
FILT=$1; shift
export TMPDIR=/tmp
cd /tmp
run() {
  tag=$1; shift
  rm -rf /tmp/pmcq_$tag
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcq_$tag -o p -- python "${CMD[@]}" > /dev/null 2>&1
  f=$(find /tmp/pmcq_$tag -name '*counter_collection.csv' | head -1)
  FILT="$FILT" python3 - "$f" <<'PY'
import csv, os, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if re.search(os.environ["FILT"], k):
        m = re.search(r"(k_\w+(<[^>]*>)?)", k)
        acc[m.group(1) if m else k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
}
CMD=("$@"); CMD[0]="$REPO/${CMD[0]}"
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SALU
run c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT

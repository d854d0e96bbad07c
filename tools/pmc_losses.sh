#!/bin/bash
# SQ / TCC counters of the mask-loss kernels (tools/bench_losses.py) -- where do k_edge_march* / k_edge_bwd_gather* wait?
#   usage (GPU box, repo root): bash tools/pmc_losses.sh
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
cd /tmp
run() {   # tag, counters...
  tag=$1; shift
  rm -rf /tmp/pmcl_$tag
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcl_$tag -o p -- python "$REPO/tools/bench_losses.py" > /dev/null 2>&1
  f=$(find /tmp/pmcl_$tag -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_edge" in k or "k_ce_fwd" in k or "k_softmax_fwd" in k:
        import re
        m = re.search(r"(k_\w+(<[^>]*>)?)", k)
        acc[m.group(1) if m else k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run b SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE
run c FETCH_SIZE
run d WRITE_SIZE
run e TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
